// Frame arithmetic of the fast iSTFT head (bf16 mode), shared by istft_head_wave_fast_kernel (kk_source.hip) and the fused
// conv_post + iSTFT kernel (kk_head.hip): spec = exp, phase = sin (istftnet.py:804-805), 11 bins -> 20 windowed samples of the inverse real
// DFT (MLXSTFT.inverse istftnet.py:497-523, istft utils.py:104-158).  Everything that does not depend on the data is a literal: twiddles
// carry the factor 2 of the real inverse DFT (exact), the output scale is 0.5 * 0.05 * hann_per[o] -- the interior window sum of a periodic
// Hann at hop N/4 is exactly 2.0f in float32 for every r in the overlap-add's summation order, so interior samples need no division.
// Compile the including file with -fno-slp-vectorize (the math is written on explicit register pairs).
#pragma once
#include "kk_common.h"

namespace kk_istft {
typedef float v2f __attribute__((ext_vector_type(2)));

// in[0..10] = log magnitudes, in[11..21] = phase arguments of one frame; fmask = 1 for an existing frame, 0 otherwise (all 20 samples are
// then exact zeros); y[o] = sample o of the frame, already multiplied by the window and the interior normalisation
__device__ __forceinline__ void frame_fast(const float (&in)[22], float fmask, float (&y)[20]) {
  constexpr float CS2[20] = {2.0f, 1.9021130800247192f, 1.6180340051651f, 1.1755704879760742f, 0.6180340051651001f, 0.0f, -0.6180340051651001f,
                             -1.1755704879760742f, -1.6180340051651f, -1.9021130800247192f, -2.0f, -1.9021130800247192f, -1.6180340051651f,
                             -1.1755704879760742f, -0.6180340051651001f, 0.0f, 0.6180340051651001f, 1.1755704879760742f, 1.6180340051651f,
                             1.9021130800247192f};
  constexpr float SN2[20] = {0.0f, 0.6180340051651001f, 1.1755704879760742f, 1.6180340051651f, 1.9021130800247192f, 2.0f, 1.9021130800247192f,
                             1.6180340051651f, 1.1755704879760742f, 0.6180340051651001f, 0.0f, -0.6180340051651001f, -1.1755704879760742f,
                             -1.6180340051651f, -1.9021130800247192f, -2.0f, -1.9021130800247192f, -1.6180340051651f, -1.1755704879760742f,
                             -0.6180340051651001f};
  // 0.5 * (0.05f * hann_per[o])
  constexpr float KH[20] = {0.0f, 0.5f * 0.0012235870817676187f, 0.5f * 0.004774575587362051f, 0.5f * 0.010305369272828102f,
                            0.5f * 0.01727457530796528f, 0.5f * 0.02500000037252903f, 0.5f * 0.03272542357444763f, 0.5f * 0.03969463333487511f,
                            0.5f * 0.04522542282938957f, 0.5f * 0.04877641424536705f, 0.5f * 0.05000000074505806f, 0.5f * 0.04877641424536705f,
                            0.5f * 0.04522542282938957f, 0.5f * 0.03969463333487511f, 0.5f * 0.03272542357444763f, 0.5f * 0.02500000037252903f,
                            0.5f * 0.01727457530796528f, 0.5f * 0.010305369272828102f, 0.5f * 0.004774575587362051f, 0.5f * 0.0012235870817676187f};
  // bins in PAIRS (2p, 2p+1): every stage below is packed fp32 math on register pairs, and the pair is exactly the (even k, odd k)
  // split the o <-> 10-o symmetry of the inverse DFT needs, so no value ever has to be moved into a pair.  Bin 11 does not exist:
  // its slot carries zeros and zero twiddles.
  v2f re2[6], im2[6];
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const v2f lm = {in[2 * p], p < 5 ? in[2 * p + 1] : 0.f};
    const v2f pr = {in[11 + 2 * p], p < 5 ? in[12 + 2 * p] : 0.f};
    const v2f mag = v2f{__expf(lm.x), __expf(lm.y)} * v2f{fmask, fmask};
    const v2f ph = {__sinf(pr.x), __sinf(pr.y)};
    // cos(ph), sin(ph) / ph for |ph| <= 1: Taylor to ph^10 (truncation < 3e-9), Horner in t = ph^2
    const v2f t = ph * ph;
    v2f c = v2f{-2.7557319223985888e-07f, -2.7557319223985888e-07f};
    c = __builtin_elementwise_fma(c, t, v2f{2.48015873015873e-05f, 2.48015873015873e-05f});
    c = __builtin_elementwise_fma(c, t, v2f{-1.3888888888888889e-03f, -1.3888888888888889e-03f});
    c = __builtin_elementwise_fma(c, t, v2f{4.1666666666666664e-02f, 4.1666666666666664e-02f});
    c = __builtin_elementwise_fma(c, t, v2f{-0.5f, -0.5f});
    c = __builtin_elementwise_fma(c, t, v2f{1.0f, 1.0f});
    v2f sn = v2f{-2.505210838544172e-08f, -2.505210838544172e-08f};
    sn = __builtin_elementwise_fma(sn, t, v2f{2.7557319223985893e-06f, 2.7557319223985893e-06f});
    sn = __builtin_elementwise_fma(sn, t, v2f{-1.984126984126984e-04f, -1.984126984126984e-04f});
    sn = __builtin_elementwise_fma(sn, t, v2f{8.333333333333333e-03f, 8.333333333333333e-03f});
    sn = __builtin_elementwise_fma(sn, t, v2f{-1.6666666666666666e-01f, -1.6666666666666666e-01f});
    sn = __builtin_elementwise_fma(sn, t, v2f{1.0f, 1.0f});
    re2[p] = mag * c;
    im2[p] = (mag * ph) * sn;
  }
  // x[o] = sum_k w_k (re_k cos(2 pi k o / 20) - im_k sin(2 pi k o / 20)), w = 1 for DC / Nyquist (whose imaginary parts the inverse
  // real FFT ignores: zero twiddles), 2 otherwise; (Ce, Co) / (Se, So) = the even-k / odd-k partial sums = the two halves of one
  // packed accumulator; outputs o, 20-o, 10-o, 10+o share them.
#pragma unroll
  for (int o = 0; o <= 5; ++o) {
    v2f Cp = {0.f, 0.f}, Sp = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int k0 = 2 * p, k1 = 2 * p + 1;
      const v2f cc = {k0 == 0 ? 1.0f : (k0 == 10 ? ((o & 1) ? -1.0f : 1.0f) : CS2[(k0 * o) % 20]), k1 > 10 ? 0.0f : CS2[(k1 * o) % 20]};
      Cp = __builtin_elementwise_fma(re2[p], cc, Cp);
      if (p < 5 && o > 0) {  // o = 0: every sine twiddle is zero
        const v2f ss = {k0 == 0 ? 0.0f : SN2[(k0 * o) % 20], SN2[(k1 * o) % 20]};
        Sp = __builtin_elementwise_fma(im2[p], ss, Sp);
      }
    }
    {
      const float C = Cp.x + Cp.y, S = Sp.x + Sp.y;
      y[o] = (C - S) * KH[o];
      if (o > 0) y[20 - o] = (C + S) * KH[20 - o];
    }
    if (o < 5) {
      const int p = 10 - o;
      const float C = Cp.x - Cp.y, S = Sp.y - Sp.x;
      y[p] = (C - S) * KH[p];
      if (p < 10) y[20 - p] = (C + S) * KH[20 - p];
    }
  }
}
}  // namespace kk_istft
