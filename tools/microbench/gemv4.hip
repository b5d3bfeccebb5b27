// Launcher of the 2-D split-K decode GEMV (gemv4_kernel.h).
#include "gemv4_kernel.h"

template <int KS, int S, int T, int EPI, bool NORM>
static int launch4(const Gemv4Args& a, int64_t n_tiles, hipStream_t s) {
  constexpr int TB = G4_WAVES * T;
  if (n_tiles % TB) return VAURA_ERR_SHAPE;
  const int64_t nb = n_tiles / TB;
  if (nb > 64) return VAURA_ERR_SHAPE;   // counters per slot
  if ((size_t)n_tiles * S * 1024 > a.scratch_bytes) return VAURA_ERR_ARG;
  VA_LAUNCH((gemv4_kernel<KS, S, T, EPI, NORM>), dim3((unsigned)(nb * S)), dim3(G4_WAVES * 64), 0, s, a);
  return 0;
}

int va_launch_gemv4(const Gemv4Args& a, int64_t n_weight_rows, int64_t K, int epilogue, bool norm, hipStream_t s) {
  if (!a.W || !a.XP || !a.scratch || !a.counters || !a.timeout || a.rows <= 0 || a.rows > 16 || (n_weight_rows % 16))
    return VAURA_ERR_ARG;
  if (norm && (!a.ss_in || a.n_ss_in <= 0)) return VAURA_ERR_ARG;
  const int64_t tiles = n_weight_rows / 16;
  if (K == 1536) {
    if (epilogue == E3_STORE && norm) return launch4<6, 8, 1, E3_STORE, true>(a, tiles, s);
    if (epilogue == E3_STORE && !norm) return launch4<6, 8, 1, E3_STORE, false>(a, tiles, s);
    if (epilogue == E3_RESID && !norm) return launch4<3, 16, 1, E3_RESID, false>(a, tiles, s);
    if (epilogue == E3_SWIGLU && norm) return launch4<6, 8, 2, E3_SWIGLU, true>(a, tiles, s);
    if (epilogue == E3_LOGITS && norm) return launch4<6, 8, 2, E3_LOGITS, true>(a, tiles, s);
  } else if (K == 4096) {
    if (epilogue == E3_RESID && !norm) return launch4<8, 16, 1, E3_RESID, false>(a, tiles, s);
    if (epilogue == E3_STORE && !norm) return launch4<8, 16, 1, E3_STORE, false>(a, tiles, s);
  }
  return VAURA_ERR_SHAPE;
}
