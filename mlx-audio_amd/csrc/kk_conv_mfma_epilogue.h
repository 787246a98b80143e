// Epilogue of the MFMA convolution kernels -- TEXTUAL INCLUDE at the end of the kernel body of kk_conv_mfma.hip (variant 2) and
// kk_conv_mfma4.hip (variant 4): accumulators -> fp32 LDS tile (one wave row group per pass) -> coalesced 16-byte rows with bias /
// activation / residual / scale / accumulate / length mask, per-tile column statistics of the stored values.  ONE copy for both kernels.
// Names taken from the including scope: a, b, bx (row-tile index), phase, nphase, q0, n0, tid, lane, wave, wr, wc, acc, Cs, red (via smem),
// tile_live, Lout, TO, WM, MI, BM, G, NRM, v2f / fma2 / gelu_exact (kk_conv_mfma_shared.h).
  // ---- epilogue: per 128 rows, accumulators -> fp32 LDS tile -> coalesced rows --------------------------------------
  TO* ob = (TO*)a.out + (long long)b * a.obs;
  const TO* rb = a.res ? (const TO*)a.res + (long long)b * a.rbs : nullptr;
  const int n = n0 + (tid & 15) * 8;  // this thread's 8 output channels (same for all its rows)
  const int nc = n < a.Cout ? n : 0;  // clamped for the unconditional loads
  const int lo_hi = a.Lo_rows - 1;
  constexpr int VEC = sizeof(TO) == 2 ? 1 : 2;  // 16-byte vectors per 8 outputs
  // every row of the tile lies inside the utterance (uniform; true for all but the last tile of an utterance)
  const bool tile_full = tile_live && !a.flat_T && (a.mode == KK_CONV ? q0 + BM - 1 : phase + a.stride * (q0 + BM - 1)) < Lout;
  v2f bias2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)  // bias has CoutP entries
    bias2[k] = a.bias ? v2f{a.bias[n0 + (tid & 15) * 8 + 2 * k], a.bias[n0 + (tid & 15) * 8 + 2 * k + 1]} : v2f{0.f, 0.f};
  const v2f scale2 = {a.scale, a.scale}, act_slope2 = {a.act_slope, a.act_slope};
  v2f st_s[4], st_q[4];  // column sums / sums of squares of the values this thread stores (pairs of adjacent columns)
#pragma unroll
  for (int k = 0; k < 4; ++k) st_s[k] = st_q[k] = v2f{0.f, 0.f};

  constexpr int RPP = G::RPP, NPASS = BM / RPP, TASKS = RPP * 16 / 256, TG = TASKS / 2;  // 8 or 6 row tasks per thread and pass
  // bf16 residual rows of the WHOLE tile are requested up front (xreg / w registers are dead by now): one exposed HBM
  // round trip per tile instead of one per (pass, half)
  constexpr bool PRE = sizeof(TO) == 2;
  uint4 rpre[PRE ? NPASS * 2 * TG : 1];
  if (PRE && rb) {
#pragma unroll
    for (int j = 0; j < NPASS * 2 * TG; ++j) {
      const int q = q0 + (j / (2 * TG)) * RPP + (((j % (2 * TG)) * 256 + tid) >> 4);
      const int op = a.mode == KK_CONV ? q : phase + a.stride * q;
      const int opc = op < 0 ? 0 : (op > lo_hi ? lo_hi : op);
      rpre[j] = *(const uint4*)(rb + (long long)opc * a.ldr + nc);
    }
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if (tile_live) {
      if (pass > 0) __syncthreads();  // previous pass's readers are done with Cs
      // rows [RPP*pass, RPP*pass + RPP) of the block tile: tall tiles -> wave row `pass`; 128-row tile -> both wave rows
      if (wr == pass) {
#ifdef KK_EPI_ACC16
        // 16 x 16 blocks (v_mfma_f32_16x16x32_bf16): lane l holds rows 4 * (l / 16) .. + 3 of column l % 16
#pragma unroll
        for (int mi = 0; mi < WM / 16; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const int col = wc * 64 + ni * 16 + (lane & 15);
            const int rbase = mi * 16 + 4 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(rbase + r) * CLD + col] = acc[mi][ni][r];
          }
#else
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int col = wc * 64 + ni * 32 + (lane & 31);
            const int rbase = mi * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) Cs[(rbase + (r & 3) + 8 * (r >> 2)) * CLD + col] = acc[mi][ni][r];
          }
#endif
      }
      __syncthreads();
    }
    if (pass == 0) TR_ADD(6, TR_NOW() - tr0);  // .. first accumulator tile is in LDS (residual requests issued)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      int opv[TG];
      bool wr_ok[TG], live[TG];
      uint4 rres[TG][VEC], rold[TG][VEC];
#pragma unroll
      for (int i = 0; i < TG; ++i) {
        const int row = ((half * TG + i) * 256 + tid) >> 4;
        const int q = q0 + pass * RPP + row;
        const int op = a.mode == KK_CONV ? q : phase + a.stride * q;
        opv[i] = op < 0 ? 0 : (op > lo_hi ? lo_hi : op);
        wr_ok[i] = q < a.Q && op < a.Lo_rows && n < a.Cout;
        live[i] = tile_live && op < Lout;
        if (a.flat_T) {  // (uniform) flat rows: the item this row belongs to decides
          const int bb = opv[i] / a.flat_T;
          live[i] = live[i] && (opv[i] - bb * a.flat_T) < kk_len(a.flat_len, bb);
        }
      }
      if (rb) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < TG; ++i) {
          if (PRE) rres[i][0] = rpre[(pass * 2 + half) * TG + i];
          else
#pragma unroll
            for (int v = 0; v < VEC; ++v) rres[i][v] = *((const uint4*)(rb + (long long)opv[i] * a.ldr + nc) + v);
        }
      }
      if (a.accumulate) {
#pragma unroll
        for (int i = 0; i < TG; ++i)
#pragma unroll
          for (int v = 0; v < VEC; ++v) rold[i][v] = *((const uint4*)(ob + (long long)opv[i] * a.ldo + nc) + v);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < TG; ++i) {
        // the element-wise chain runs on float PAIRS (v_pk_add/mul/fma_f32: two lanes of fp32 per instruction, same IEEE
        // results as the scalar form); bf16 <-> fp32 widening is a shift / mask on the packed word
        const int row = ((half * TG + i) * 256 + tid) >> 4;
        v2f v[4];
        if (tile_live) {
          const float4 c0 = *(const float4*)(Cs + row * CLD + (tid & 15) * 8);
          const float4 c1 = *(const float4*)(Cs + row * CLD + (tid & 15) * 8 + 4);
          v[0] = v2f{c0.x, c0.y}; v[1] = v2f{c0.z, c0.w}; v[2] = v2f{c1.x, c1.y}; v[3] = v2f{c1.z, c1.w};
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v2f{0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += bias2[k];
        if (a.act == KK_ACT_LRELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const v2f m = v[k] * act_slope2;
            v[k].x = v[k].x > 0.f ? v[k].x : m.x;
            v[k].y = v[k].y > 0.f ? v[k].y : m.y;
          }
        } else if (a.act == KK_ACT_GELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k].x = gelu_exact(v[k].x);
            v[k].y = gelu_exact(v[k].y);
          }
        } else if (NRM == 0 && a.act == KK_ACT_GELU_TANH) {  // nn.gelu_approx (plain variant only: keeps the fused variants' registers)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k].x = 0.5f * v[k].x * (1.0f + tanhf(0.7978845608028654f * (v[k].x + 0.044715f * (v[k].x * v[k].x * v[k].x))));
            v[k].y = 0.5f * v[k].y * (1.0f + tanhf(0.7978845608028654f * (v[k].y + 0.044715f * (v[k].y * v[k].y * v[k].y))));
          }
        }
        if (rb) {
          if (sizeof(TO) == 2) {
            const unsigned w4[4] = {rres[i][0].x, rres[i][0].y, rres[i][0].z, rres[i][0].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += v2f{__uint_as_float(w4[k] << 16), __uint_as_float(w4[k] & 0xFFFF0000u)};
          } else {
            U32x8 t;
            t.u[0] = rres[i][0];
            t.u[1] = rres[i][VEC - 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += v2f{t.f[2 * k], t.f[2 * k + 1]};
          }
        }
        if (a.scale != 1.0f) {  // (uniform; 1 everywhere but the mean over the three resblocks)
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] *= scale2;
        }
        if (a.accumulate) {
          if (sizeof(TO) == 2) {
            const unsigned w4[4] = {rold[i][0].x, rold[i][0].y, rold[i][0].z, rold[i][0].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += v2f{__uint_as_float(w4[k] << 16), __uint_as_float(w4[k] & 0xFFFF0000u)};
          } else {
            U32x8 t;
            t.u[0] = rold[i][0];
            t.u[1] = rold[i][VEC - 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += v2f{t.f[2 * k], t.f[2 * k + 1]};
          }
        }
        if (a.post_slope != 0.f && a.post_slope != 1.0f) {  // (uniform) LeakyReLU of the finished value: max(v, slope v) for 0 < slope < 1
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float mx = v[k].x * a.post_slope, my = v[k].y * a.post_slope;
            v[k].x = v[k].x > mx ? v[k].x : mx;
            v[k].y = v[k].y > my ? v[k].y : my;
          }
        }
        if (sizeof(TO) == 2) {
          // rows past the utterance are stored as exact zeros; only a tile that reaches past it pays the selects (tile_full is uniform)
          if (!tile_full && !live[i]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v2f{0.f, 0.f};
          }
          unsigned w4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bf16x2 pk = {(bf16_t)v[k].x, (bf16_t)v[k].y};
            w4[k] = __builtin_bit_cast(unsigned, pk);
          }
          if (wr_ok[i]) {
            *(uint4*)(ob + (long long)opv[i] * a.ldo + n) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            if (a.stat_part) {
              // column statistics for the next instance norm, from the fp32 values BEFORE the bf16 rounding (round 3: the consumer reads the
              // rounded tensor, whose sums differ from these by the rounding noise averaged over the rows, ~1e-5 relative; the fp32 oracle
              // normalises unrounded values too; re-widening the packed words cost 2 of the 8 VALU instructions per element pair here)
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                st_s[k] += v[k];
                st_q[k] = fma2(v[k], v[k], st_q[k]);
              }
            }
          }
        } else {
          if (!live[i]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v2f{0.f, 0.f};
          }
          if (wr_ok[i]) {
            float* dst = (float*)(ob + (long long)opv[i] * a.ldo + n);
            *(float4*)dst = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
            *(float4*)(dst + 4) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
          }
        }
      }
    }
    if (pass == 0) TR_ADD(7, TR_NOW() - tr0);  // .. first pass stored
  }
  TR_ADD(4, TR_NOW() - tr0);  // start .. end of the store phase
  TR_ADD(5, 1);
  if (a.stat_part) {
    // rows of one column group live in threads tid = rg*16 + cg: reduce rg over the wave by shuffles (xor 16, 32),
    // then over the 4 waves through LDS; one deterministic partial per (utterance, tile, column)
    __syncthreads();  // every wave is done reading the Cs tile
    float* red = (float*)smem;  // [4 waves][2][128]
    float ss[8], sq[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ss[2 * k] = st_s[k].x; ss[2 * k + 1] = st_s[k].y;
      sq[2 * k] = st_q[k].x; sq[2 * k + 1] = st_q[k].y;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ss[k] += __shfl_xor(ss[k], 16);
      ss[k] += __shfl_xor(ss[k], 32);
      sq[k] += __shfl_xor(sq[k], 16);
      sq[k] += __shfl_xor(sq[k], 32);
    }
    if (lane < 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(wave * 2 + 0) * 128 + lane * 8 + k] = ss[k];
        red[(wave * 2 + 1) * 128 + lane * 8 + k] = sq[k];
      }
    }
    __syncthreads();
    const int which = tid >> 7, col = tid & 127;  // threads 0..127 -> sums, 128..255 -> sums of squares
    if (n0 + col < a.Cout) {
      const float v = red[(0 * 2 + which) * 128 + col] + red[(1 * 2 + which) * 128 + col] + red[(2 * 2 + which) * 128 + col] +
                      red[(3 * 2 + which) * 128 + col];
      const int tile = bx * nphase + phase;
      a.stat_part[(((long long)b * a.stat_ntiles + tile) * 2 + which) * a.Cout + n0 + col] = v;
    }
  }
