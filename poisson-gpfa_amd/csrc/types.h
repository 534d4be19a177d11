// Types and small device helpers shared by the kernel headers and by every translation unit of the library (ctx.h):
// GEMM launch descriptor + its device-side decoding helpers (kernels: gemm.h), factor workspace (kernels: chol.h),
// control block of the inner PCG loop (kernels: pcg.h).  No kernels here: a translation unit only compiles the kernels of the
// headers it includes.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace pgpfa {

typedef double double4_t __attribute__((ext_vector_type(4)));

enum : int {
  GEMM_FULL = 0,        // all tilesM x tilesN tiles
  GEMM_LOWER = 1,       // only tiles with ti >= tj (C, row/col origins coincide)
};
enum : int {
  KF_BEGIN_ROW = 1,     // A is upper-triangular in tile units: k starts at ti*128
  KF_END_ROW = 2,       // A is lower-triangular in tile units: k ends at (ti+1)*128
  KF_MASK_DIAG = 4,     // on tiles with ti == tj store only i >= j
  KF_BEGIN_MAXRC = 8,   // both operands upper-triangular: k starts at max(ti,tj)*128
};

struct GemmP {
  const double* A; long long sA; int lda;
  const double* B; long long sB; int ldb;
  double* C; long long sC; int ldc;
  int M, N, K;
  double alpha, beta;
  const int* slots;     // batch b -> slot (NULL: identity); pointers advance by slot*stride
  // optional row-tile table [ntab][4] = {first row, end row, k begin, k end} (k multiples of 16, first row even): the row tiles of a block-
  // sparse A (block-diagonal factors: one latent per tile, so that no tile straddles two latents' zeros).  Tile ti covers rows
  // [first, min(first + bm, end)); NULL: uniform tiles of bm rows and the kflags rule
  const int* rtab; int ntab;
  int bm;               // tile size (rows = columns of a workgroup tile): 128, or 64 for products with few tiles (set by gemm_launch)
  int nbatch;
  int mode, kflags;
  int tilesM, tilesN, ntiles;
  double flops_hint;    // algorithmic flops of the launch when the operands are block sparse (0: dense formula)
  int k_loop_hint;      // longest k loop of a tile when rtab is set (0: derive it from flops_hint / K); host-side use only
  // optional two-level batch: entry b = hi * nb_lo + lo; `slots` maps lo, hi adds its own strides (nb_lo = 0: one level)
  int nb_lo;
  long long sA_hi, sB_hi, sC_hi;
  // optional segmented K (TRANSB = 0 only): column k of A/B lives at (k / kseg) * s?seg + (k % kseg) * ld, i.e. the K
  // dimension runs over kseg-wide panels of consecutive slabs (kseg multiple of 16; 0: plain)
  int kseg;
  long long sAseg, sBseg;
  // optional split-K: with ksplit > 1 the hi index of the two-level batch selects the ksplit-th part of the tile's k
  // range instead of moving A/B (sA_hi = sB_hi = 0); C then addresses partial products (gemm_splitk_reduce_kernel)
  int ksplit;
  int c_by_pos;         // C is indexed by batch position instead of slot (compact partial-product buffers)
  const int* skip;      // optional device flag: the launch is a no-op when *skip != 0 (device-side loop control, pcg.h)
  // optional column list (TRANSB = 1 only): column j of the product is column cols[j] of B and of C, j < N - the multi-RHS products of
  // the Newton-PCG run over the LIVE slots only, wherever those sit among the chunk's slot vectors
  const int* cols;
  int cols_c_off;       // the column list applies to B only (C is a compact partial-product buffer: split-K)
  const int* n_dev;     // optional: the number of columns is *n_dev (<= N) - the length of a device-side list; tiles past it return
  int b_f32;            // the B operand is stored in single precision (NT form, 64 x 64 tiles, MFMA path only)
};

__device__ __forceinline__ size_t gemm_koff(int kseg, long long sseg, int ld, int k) {
  if (kseg == 0) return (size_t)k * ld;
  const int seg = k / kseg;
  return (size_t)seg * sseg + (size_t)(k - seg * kseg) * ld;
}

__device__ __forceinline__ int gemm_bind(const GemmP& g, int b, const double*& A, const double*& B, double*& C) {
  int lo = b, hi = 0;
  if (g.nb_lo > 0) { hi = b / g.nb_lo; lo = b - hi * g.nb_lo; }
  const long long slot = g.slots ? g.slots[lo] : lo;
  A = g.A + slot * g.sA + hi * g.sA_hi;
  B = g.B + slot * g.sB + hi * g.sB_hi;
  C = g.C + (g.c_by_pos ? (long long)lo : slot) * g.sC + hi * g.sC_hi;
  return hi;
}

// k range of one split-K part (multiples of 16)
__device__ __forceinline__ void gemm_split_range(int ksplit, int part, int& kb, int& ke) {
  if (ksplit <= 1) return;
  const int len = ke - kb;
  const int chunk = ((len + ksplit - 1) / ksplit + 15) / 16 * 16;
  const int b0 = kb + part * chunk;
  ke = (b0 + chunk < ke) ? b0 + chunk : ke;
  kb = (b0 < ke) ? b0 : ke;
}

__device__ __forceinline__ void gemm_decode_tile(const GemmP& g, int tile, int& ti, int& tj) {
  if (g.mode == GEMM_FULL) {
    ti = tile % g.tilesM;
    tj = tile / g.tilesM;
  } else {
    // column tj holds tilesM - tj tiles (ti = tj .. tilesM-1)
    int tjj = 0, rem = tile;
    while (rem >= g.tilesM - tjj) { rem -= g.tilesM - tjj; ++tjj; }
    tj = tjj;
    ti = tjj + rem;
  }
}

// Block id -> (batch entry, tile).  The hardware deals consecutive block ids round-robin over the 8
// XCDs (each with a private L2), so batch entries are grouped by 8: XCD x walks the tiles of entry
// 8g+x in order.  The ~64 blocks resident on one XCD then belong to one trial and share its A/B panels
// in that XCD's L2 (and the 256 MB Infinity Cache), instead of every block streaming private panels
// from HBM.  Placement only affects speed, never results.
__device__ __forceinline__ void gemm_decode_block(const GemmP& g, int bid, int& b, int& tile) {
  const int full = g.nbatch >> 3;                 // complete groups of 8 entries
  const int per_group = g.ntiles << 3;
  const int grp = bid / per_group;
  if (grp < full) {
    const int r = bid - grp * per_group;
    tile = r >> 3;
    b = (grp << 3) + (r & 7);
  } else {
    const int m = g.nbatch - (full << 3);         // remainder group of m < 8 entries
    const int r = bid - full * per_group;
    tile = r / m;
    b = (full << 3) + (r - tile * m);
  }
}

constexpr int GBM = 128, GBN = 128, GBK = 16;   // the large tile (GBN is also the column-tile unit of the zero-skipping consumers of Yt)

// element-type helpers of the MFMA kernel: FP64 (the E-step) and FP32 (mixed-precision dual-variational evaluation: same
// 16x16x4 tile shape and fragment layout, v_mfma_f32_16x16x4_f32 issues at twice the FP64 rate)
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
template <typename T> struct GemmVec;
template <> struct GemmVec<double> { using v2 = double2_t; using v4 = double4_t; };
template <> struct GemmVec<float> { using v2 = float2_t; using v4 = float4_t; };
__device__ __forceinline__ double4_t gemm_mfma16(double a, double b, double4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float4_t gemm_mfma16(float a, float b, float4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// operand pointers of batch entry b for element type T (GemmP carries them as double*; strides count elements of T)
template <typename T>
__device__ __forceinline__ int gemm_bind_t(const GemmP& g, int b, const T*& A, const T*& B, T*& C) {
  int lo = b, hi = 0;
  if (g.nb_lo > 0) { hi = b / g.nb_lo; lo = b - hi * g.nb_lo; }
  const long long slot = g.slots ? g.slots[lo] : lo;
  A = reinterpret_cast<const T*>(g.A) + slot * g.sA + hi * g.sA_hi;
  B = reinterpret_cast<const T*>(g.B) + slot * g.sB + hi * g.sB_hi;
  C = reinterpret_cast<T*>(g.C) + (g.c_by_pos ? (long long)lo : slot) * g.sC + hi * g.sC_hi;
  return hi;
}

constexpr int NB = 128;     // diagonal block / GEMM tile
constexpr int NSUP = 512;   // super-panel width

struct CholWS {
  double* H; long long sH;        // factor slabs
  double* Mt; long long sM;       // L^-T slabs (strictly-lower part must be zero)
  double* Dinv; long long sD;     // inverted diagonal blocks, (npad/128) x 128 x 128 per slot
  double* P; long long sP;        // npad x 128 scratch per slot
  int* info;                      // per slot
  int ld, npad;
  int nact = 0;                   // active rows (multiple of 64, <= npad); 0 = npad
};

struct PcgCtl {
  int stop; int iters; unsigned worst_bits; int nlive; unsigned long long slot_iters;
  int nl[2];            // two-kernel step (pcg_cg_a/b_kernel): lengths of the two live lists (this step's, the next one's)
  int pad_[2];
};

// launch of the GEMM kernel (definition: gemm.h, compiled in linalg.hip)
hipError_t gemm_launch(hipStream_t st, bool use_mfma, bool transb, GemmP g, bool f32 = false);

}  // namespace pgpfa
