// Parameter block of the NHWC convolution kernels (gemm_conv.hip: register-staged tiles, LDS-resident 3x3 / 64-channel
// kernel; conv_ring.hip: the register-ring kernels; pp8.hip: the eight-wave 128x128 kernel).  Host-built, passed by value.
#pragma once

namespace drn_conv {

struct ConvParams {
  const char* X;
  const char* Wt;   // [Cout][ldw] K-major, k = (kh*KW + kw)*Cin + ci
  char* Y;          // [Nb*Ho*Wo][ldy]
  const float* scale;  // per Cout (FrozenBN folded) or null => 1
  const float* bias;   // per Cout or null => 0
  const char* residual;  // same layout as Y, or null
  int Nb, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil, relu;
  int Ktot;  // KH*KW*Cin
  long ldw, ldy, ldres;
  // quantised trunk (drn_conv2d_nhwc_q): the output / residual element types may differ from the input's (bf16 stem ->
  // fp8, fp8 -> bf16 feature map), and an fp8 residual carries its own per-tensor scale (res_mult = s_out / s_res)
  int out_dt, res_dt;
  float res_mult;
  int fp8_k64;  // fp8 operands: 1 = the K = 64 scaled MFMA (fp8 rate), 0 = the K = 16 form (DRN_TUNE_FP8_K64)
  // conv3x3_c64_kernel<.., PW = true> (drn_conv3x3_pw_nhwc): a 1x1 convolution 64 -> 256 channels on this conv's output,
  // which never leaves the chip - the tail of a res2 bottleneck.  Y / ldy / residual / ldres / res_mult then belong to THAT
  // layer (256 channels), scale / bias / relu above to the 3x3, pw_* to the 1x1
  const char* pw_w; long pw_ldw;  // [256][pw_ldw] K-major (64 input channels)
  const float* pw_scale; const float* pw_bias;
  int pw_relu;
  // conv3x3_c64_kernel<.., POOL = true>: nn.MaxPool2d(2, 2) on the (ReLU'd) output, in the epilogue - Y is the POOLED map
  // [Nb][(Ho - 2) / 2 + 1][(Wo - 2) / 2 + 1][channels]; the full-resolution output is never written
  int pool;
};

}  // namespace drn_conv
