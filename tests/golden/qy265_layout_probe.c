#include <stddef.h>
#include <stdio.h>
#include HDR
#define O(f) printf("\"%s\": %zu,\n", #f, offsetof(QY265EncConfig, f))
int main(void){
 printf("{\n");
 O(pAuth);O(tune);O(preset);O(latency);O(profileId);O(bHeaderBeforeKeyframe);O(picWidth);O(picHeight);O(frameRate);O(bframes);O(temporalLayer);O(vpp_denoise);O(vpp_hdr_strength);O(vpp_recur_filter);
 O(rc);O(bitrateInkbps);O(vbv_buffer_size);O(qp);O(crf);O(iIntraPeriod);O(qpmin);O(qpmax);O(enFrameSkip);O(enWavefront);O(enFrameParallel);O(threads);O(vui_parameters_present_flag);O(vui.matrix_coeffs);
 O(logLevel);O(lookahead);O(calcPsnr);O(calcSsim);O(shortLoadingForPlayer);O(iPass);O(statFileName);O(fRateTolerance);O(rdoq);O(me);O(part);O(do64);O(tuInter);O(tuIntra);O(smooth);O(transskip);O(subme);
 O(satdInter);O(satdIntra);O(searchrange);O(refnum);O(ref0);O(sao);O(longTermRef);O(iAqMode);O(fAqStrength);O(rasl);
 printf("\"sizeof_config\": %zu, \"sizeof_yuv\": %zu, \"sizeof_picture\": %zu, \"sizeof_nal\": %zu,\n", sizeof(QY265EncConfig), sizeof(QY265YUV), sizeof(QY265Picture), sizeof(QY265Nal));
 printf("\"nal_payload\": %zu, \"pic_yuv\": %zu, \"yuv_stride\": %zu\n}\n", offsetof(QY265Nal,pPayload), offsetof(QY265Picture,yuv), offsetof(QY265YUV,iStride));
 return 0; }
