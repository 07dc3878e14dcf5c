#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel of libks265hip.so: hipcc -Rpass-analysis=kernel-resource-usage over ks265codec_amd/csrc/*.hip (cross-compiles without a GPU).
usage: python tools/resource_usage.py > profiles/rNN_resource_usage.txt"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = ("intra_decide_kernel", "intra_gate_kernel", "intra_recon_kernel", "deblock_kernel", "sao_ctu_kernel<false>", "bi_decide_kernel<false, false>", "bi_refine_chosen_kernel<false>", "cu_decide_kernel", "me_subpel_kernel<8, false>",
       "merge_pass_kernel", "me_propagate_kernel", "ref_decide_kernel", "me_int_kernel", "me_order_kernel", "me_score_kernel", "presearch_", "pyr_down_kernel", "reconstruct_kernel", "cfc_", "cutree_finish_kernel", "qoff_ctu_map_kernel",
       "pack_compact_kernel", "pack_records_kernel", "copy_out_compact_kernel", "pad_picture_kernel", "sse_picture", "load_i420", "unpack")
rows = []
for src in sorted(glob.glob(os.path.join(ROOT, "ks265codec_amd", "csrc", "*.hip"))):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", out, re.S):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        rows.append((os.path.basename(src), name, int(m.group(2)), int(m.group(3)), int(m.group(5)), int(m.group(4))))
print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage over ks265codec_amd/csrc/*.hip (every kernel of libks265hip.so); * = on the path of the headline GOP / of -rc 3")
print("# scratch = bytes per lane (register spills or private arrays)")
print(f"{'kernel':62s} {'file':22s} {'VGPRs':>6s} {'scratch':>8s} {'LDS':>7s} {'waves/SIMD':>10s}")
for f, n, v, s, l, o in rows:
    print(f"{'*' if any(h in n for h in HOT) else ' '}{n:61s} {f:22s} {v:6d} {s:8d} {l:7d} {o:10d}")
