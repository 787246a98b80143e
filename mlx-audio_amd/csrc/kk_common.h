// Common device/host helpers for the Kokoro HIP library (gfx950 / MI355X only).
//
// Data layout: every activation is "frames-major, channels-last"
//   x[b][l][c]  at  base + b*bstride + l*ld + c        (ld >= C: row pitch in elements)
// so that a concat along channels is a write into a channel slice of a wider buffer, conv
// taps are whole-row offsets, and MFMA A/B fragments are 16-byte reads along c.
// Per-utterance valid lengths are device int32 arrays; a kernel derives the length in its own
// domain as len[b]*mul + add and writes ZEROS past it, which is exactly the zero padding a
// B=1 call of the reference sees (kokoro.py:135-136 runs batch 1 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;

#define KK_WAVE 64

template <typename T> __device__ __forceinline__ float kk_ld(const T* p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void kk_st(T* p, float v) { *p = (T)v; }

enum KKAct { KK_ACT_NONE = 0, KK_ACT_LRELU = 1, KK_ACT_GELU = 2, KK_ACT_SNAKE = 3, KK_ACT_GELU_TANH = 4, KK_ACT_ELU = 5 };
enum KKConvMode { KK_CONV = 0, KK_CONVT = 1 };
enum KKDType { KK_F32 = 0, KK_BF16 = 1, KK_I32 = 2, KK_F16 = 3 };

// Per-batch length in a kernel's own domain: L(b) = len ? len[b]*mul + add : add
struct KKLen {
  const int* len;
  int mul, add;
};
__host__ __device__ __forceinline__ int kk_len(const KKLen& l, int b) { return l.len ? l.len[b] * l.mul + l.add : l.add; }

// Arguments of the generic (tap-loop) convolution family; see kk_conv.hip.
struct KKConvArgs {
  const void* x;      // [B][Lin_max][ldx]
  long long xbs;      // batch stride (elements)
  int ldx;
  const float* w;     // packed [Kw][Cin][ldw] fp32, ldw >= Cout
  int ldw;
  const float* bias;  // [Cout] or null
  void* out;          // [B][Lout_max][ldo]
  long long obs;
  int ldo;
  const void* res;    // optional residual, same indexing as out (own pitch)
  long long rbs;
  int ldr;
  int Cin, Cout, Kw;
  int mode;           // KK_CONV / KK_CONVT
  int stride, pad, dil;
  int in_shift;       // conv only: input row = (q*stride - pad + t*dil) >> in_shift (nearest x2 up-sampling of the input)
  int Q;              // rows per phase to cover (max over batch)
  int Lo_rows;        // rows that exist in the output buffer (writes beyond are skipped)
  KKLen lin, lout;    // valid input / output rows per utterance
  float in_slope;     // leaky-relu applied to the input on load (1.0f = identity)
  float scale;        // epilogue: v = (acc + bias + res) * scale
  int accumulate;     // epilogue: v += out
  int act;            // epilogue activation (KKAct) applied to acc+bias, before residual/scale
  float act_slope;
  int in_act;         // 0: leaky-relu(in_slope) on the input (identity at 1.0); KK_ACT_ELU: elu(x, 1) on the input (Mimi SEANet)
};

__host__ __device__ static inline int kk_cdiv(int a, int b) { return (a + b - 1) / b; }

#define KK_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return kk_fail(hipGetErrorString(e__)); \
  } while (0)

int kk_fail(const char* msg);  // records the message for kk_last_error(), returns -1

// hipFuncSetAttribute state (dynamic LDS limit) is per DEVICE, not per process: one bit per device ordinal.  The call is idempotent, so two
// threads racing through first() only repeat it.  `static KKDevOnce once; if (once.first()) { hipFuncSetAttribute(...); once.done(); }`
#ifdef __cplusplus
#include <atomic>
struct KKDevOnce {
  std::atomic<unsigned long long> mask{0};
  static unsigned long long bit() {
    int d = 0;
    (void)hipGetDevice(&d);
    return 1ull << (d & 63);
  }
  bool first() const { return (mask.load(std::memory_order_acquire) & bit()) == 0; }
  void done() { mask.fetch_or(bit(), std::memory_order_release); }
};
#endif
