// X-slab staging of the MFMA convolution kernels -- TEXTUAL INCLUDE inside the kernel body (`if (tile_live) { ... }` scope) of
// kk_conv_mfma.hip (variant 2, LDS-staged W) and kk_conv_mfma4.hip (variant 4, W fragments from global memory): ONE copy of the slab
// prefetch, the fused AdaIN + Snake / LeakyReLU transform and the padding masks, so the two kernels cannot drift apart (they are compared
// bit for bit by the `mfma4` test fixture).  Names taken from the including scope: a, b, tid, q0, off0, min_off, xrows, xb, Lin, lin_hi,
// cin_real, Xs, Ps, XREG, XLD, CK, NRM, bf16_t / bf16x2.  Defines: xreg, preg, xok, load_x(chunk), store_p(chunk), store_x(chunk).
    uint4 xreg[XREG];
    float4 preg = make_float4(0.f, 0.f, 0.f, 0.f);  // lanes 0..15 of waves 0 / 1 / 2: one float4 of the slab's A / B / alpha
    unsigned xok = 0;

    auto load_x = [&](int chunk) __attribute__((always_inline)) {
      xok = 0;
      // parameter loads go FIRST: vmcnt retires in order, so storing them to LDS one tap later does not wait for the slab
      // ONE unconditional float4 per thread, its source chosen per WAVE (wave 0: A, 1: B, 2: alpha; lanes 0..15 hold the slab's 64 channels,
      // the other lanes and wave 3 re-read valid rows and store nothing): a load under `if (which == ...)` is waited for where the paths
      // merge -- vmcnt(0), i.e. also for the weight fragments in flight, once per slab -- and a per-lane choice of the base pointer becomes a
      // dependent table load.  Alpha's "1 past the real channels" is applied at store_p; alpha is read in whole float4 groups
      // (nrm_C % 4 == 0, checked by the launcher).
      if (NRM) {
        const int which = __builtin_amdgcn_readfirstlane(tid >> 6), c = chunk * CK + (tid & 15) * 4;
        const float* base = (NRM == 1 && which == 2) ? a.nrm_alpha : (which == 1 ? a.nrm_b : a.nrm_a) + (long long)b * a.nrm_stride;
        const int off = (NRM == 1 && which == 2) ? (c + 3 < a.nrm_C ? c : 0) : c;
        preg = *(const float4*)(base + off);
      }
#pragma unroll
      for (int i = 0; i < XREG; ++i) {
        const int id = i * 256 + tid;
        const int r = id >> 3, c8 = (id & 7) * 8;
        int row = q0 + min_off + r;
        const bool ok0 = row >= 0 && r < xrows;
        if (a.in_shift) row >>= a.in_shift;
        if (ok0 && row < Lin) xok |= 1u << i;
        const int rc = row < 0 ? 0 : (row > lin_hi ? lin_hi : row);
        xreg[i] = *(const uint4*)(xb + (long long)rc * a.ldx + chunk * CK + c8);
      }
      asm volatile("" ::: "memory");
    };
    auto store_p = [&](int chunk) __attribute__((always_inline)) {
      if (NRM && tid < 192 && (tid & 63) < 16) {
        const int which = tid >> 6, l = tid & 15;
        float4 p = preg;
        if (NRM == 1 && which == 2 && chunk * CK + l * 4 + 3 >= a.nrm_C) p = make_float4(1.0f, 1.0f, 1.0f, 1.0f);  // pad channels
        *(float4*)(Ps + (chunk & 1) * 3 * CK + which * CK + l * 4) = p;
      }
    };
    auto store_x = [&](int chunk) __attribute__((always_inline)) {
      // value barrier: without it hipcc hoists the first unpack instructions of this function up to the loads in load_x
      // (one k-slab earlier) and waits for the slab there, which turns the prefetch into a synchronous load
#pragma unroll
      for (int i = 0; i < XREG; ++i) asm volatile("" : "+v"(xreg[i].x), "+v"(xreg[i].y), "+v"(xreg[i].z), "+v"(xreg[i].w));
      if (NRM) {
        // y = act(x * A + B): AdaIN1d + Snake / LeakyReLU (istftnet.py:333-337,382) applied while staging.  This thread's 8
        // channels are the same for all its rows ((id & 7) == (tid & 7)); they are handled one packed PAIR at a time so that
        // only 8 parameter registers are live, and the loop body is branch-free (the activation is a template parameter) so
        // the transcendental latency of neighbouring elements overlaps.
        const float* pt = Ps + (chunk & 1) * 3 * CK + (tid & 7) * 8;
        const float slope = a.nrm_slope;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const float a0 = pt[2 * kk], a1 = pt[2 * kk + 1], b0 = pt[CK + 2 * kk], b1 = pt[CK + 2 * kk + 1];
          float l0 = 0.f, l1 = 0.f, i0 = 0.f, i1 = 0.f;
          if (NRM == 1) {  // snake: y + sin^2(alpha y) / alpha; v_sin_f32 takes revolutions
            const float al0 = pt[2 * CK + 2 * kk], al1 = pt[2 * CK + 2 * kk + 1];
            l0 = al0 * 0.15915494309189535f;
            l1 = al1 * 0.15915494309189535f;
            i0 = __builtin_amdgcn_rcpf(al0);
            i1 = __builtin_amdgcn_rcpf(al1);
          }
#pragma unroll
          for (int i = 0; i < XREG; ++i) {
            const unsigned wd = kk == 0 ? xreg[i].x : kk == 1 ? xreg[i].y : kk == 2 ? xreg[i].z : xreg[i].w;
            float y0 = __builtin_fmaf(__uint_as_float(wd << 16), a0, b0);
            float y1 = __builtin_fmaf(__uint_as_float(wd & 0xFFFF0000u), a1, b1);
            if (NRM == 1) {
              const float s0 = __builtin_amdgcn_sinf(l0 * y0), s1 = __builtin_amdgcn_sinf(l1 * y1);
              y0 = __builtin_fmaf(i0 * s0, s0, y0);
              y1 = __builtin_fmaf(i1 * s1, s1, y1);
            } else {
              y0 = y0 > 0.f ? y0 : y0 * slope;
              y1 = y1 > 0.f ? y1 : y1 * slope;
            }
            const bf16x2 pk = {(bf16_t)y0, (bf16_t)y1};
            const unsigned o = __builtin_bit_cast(unsigned, pk);
            if (kk == 0) xreg[i].x = o;
            else if (kk == 1) xreg[i].y = o;
            else if (kk == 2) xreg[i].z = o;
            else xreg[i].w = o;
          }
        }
      }
      // A fused slab that lies inside the utterance and inside the real channels has nothing to mask (uniform over the workgroup; all but
      // the edge tiles): straight to LDS, without the two ANDs per word
      if (NRM != 0) {
        const int r0 = q0 + min_off, r1 = q0 + min_off + xrows - 1;
        if (r0 >= 0 && (a.in_shift ? (r1 >> a.in_shift) : r1) < Lin && chunk * CK + CK <= cin_real) {
#pragma unroll
          for (int i = 0; i < XREG; ++i) {
            const int id = i * 256 + tid;
            const int r = id >> 3, c8 = (id & 7) * 8;
            if (r < xrows) *(uint4*)(Xs + r * XLD + c8) = xreg[i];
          }
          return;
        }
      }
      // this thread's 8 channels are the same for all its rows; channels >= Cin are pad and may hold anything (NaN x 0 = NaN)
      const int cfirst = chunk * CK + (tid & 7) * 8;
      unsigned cm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cm[j] = (cfirst + 2 * j < cin_real ? 0x0000FFFFu : 0u) | (cfirst + 2 * j + 1 < cin_real ? 0xFFFF0000u : 0u);
#pragma unroll
      for (int i = 0; i < XREG; ++i) {
        const int id = i * 256 + tid;
        const int r = id >> 3, c8 = (id & 7) * 8;
        if (r < xrows) {
          const unsigned msk = (xok >> i) & 1u ? 0xFFFFFFFFu : 0u;
          // padding rows / pad channels stay exactly zero.  32-bit integer ops only: touching the slab registers as bf16
          // ELEMENTS makes hipcc split them into 16-bit pieces right at the loads (and wait for the loads there)
          unsigned wq[4] = {xreg[i].x & msk & cm[0], xreg[i].y & msk & cm[1], xreg[i].z & msk & cm[2], xreg[i].w & msk & cm[3]};
          if (NRM == 0 && a.in_act == KK_ACT_ELU) {  // nn.elu: where(x > 0, x, exp(x) - 1); elu(0) = 0 keeps the padding zero
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float lo = __uint_as_float(wq[k] << 16), hi = __uint_as_float(wq[k] & 0xFFFF0000u);
              lo = lo > 0.f ? lo : __expf(lo) - 1.0f;
              hi = hi > 0.f ? hi : __expf(hi) - 1.0f;
              const bf16x2 pk = {(bf16_t)lo, (bf16_t)hi};
              wq[k] = __builtin_bit_cast(unsigned, pk);
            }
          } else if (a.in_slope != 1.0f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float lo = __uint_as_float(wq[k] << 16), hi = __uint_as_float(wq[k] & 0xFFFF0000u);
              lo = lo > 0.f ? lo : lo * a.in_slope;
              hi = hi > 0.f ? hi : hi * a.in_slope;
              const bf16x2 pk = {(bf16_t)lo, (bf16_t)hi};
              wq[k] = __builtin_bit_cast(unsigned, pk);
            }
          }
          *(uint4*)(Xs + r * XLD + c8) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
        }
      }
    };
