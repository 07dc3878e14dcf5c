cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3fix
./ks265codec_amd/ks265enc -i scratch/fix.yuv -wdt 416 -hgt 240 -fr 50 -preset slow -rc 0 -qp 27 -iper 128 -bframes 0 -threads 3 -psnr 2 -b /dev/shm/f.265 > gpurun_out/r3fix/log.txt 2>&1
echo "expected 719933c7f1ec41abe1a7ab400b3cbae0 17925 bytes; got $(md5sum < /dev/shm/f.265 | cut -c1-32) $(stat -c %s /dev/shm/f.265) bytes" | tee gpurun_out/r3fix/md5.txt
grep -P "^\d+\t[IPB]" gpurun_out/r3fix/log.txt | cut -f1,2,7 | tr '\n' ' '
