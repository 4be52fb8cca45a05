// Fused SHMP layer for gfx950: SAGEConv gather-aggregate for every relation slot, the per-relation
// Linear, the to_hetero sum, the update Linear and the ReLU (gnn_model.py:262-264, 273, 389-395) in
// ONE launch; the aggregates never touch HBM.
//
//   out[i] = relu( sum_{s<sm} (sum_{e in vrow(i,s)} x[vcol[e]]) * Wt[s] + x[i] * Wt[sm] + bias
//                  + sum_{sm<=s<sm+st} sum_{e in vrow(i,s)} ytab[vcol[e]-ytab_row0][(s-sm)*64 : +64] )
//
// "MFMA slots" (s < sm) are gathered into LDS and multiplied on the matrix cores; "table slots"
// are the same linear map re-associated, (sum_j x_j) W = sum_j (x_j W): their sources were already
// multiplied by W (ytab), so they are a row add in the epilogue.  The host uses the table form for
// the canonical->count relations (at most one source per row), cutting K from 320 to 192.
//
// Structure (wave-autonomous, persistent, software-pipelined):
//   * block = 12 waves = one CU (3 waves per SIMD); all (sm+1) 64x64 weight blocks are loaded into
//     LDS ONCE per block and stay resident while the block strides over 384-row tiles;
//   * every wave owns 32 destination rows of a tile end to end and never meets a block barrier in
//     the tile loop: while two waves of a SIMD wait on gathers the third feeds the MFMA pipe;
//   * the CSR slice of the wave's NEXT tile (row pointers, then up to WCAP source ids) is
//     prefetched into a second private LDS buffer under the current tile's work, so the only
//     dependent global access on the critical path is the feature row of a neighbour;
//   * a K block is processed as two 32-column halves: 8-lane groups x float4 gather one 128-B
//     half row per neighbour (8 rows in flight per pass, 4 passes kept in flight together) into a
//     private [32][33] A image, then 32 MFMAs (32x64 output, two accumulators share the A fragment);
//   * rows with more than 4 sources in a slot (hub rows, canonical rows of dense neighborhoods)
//     are finished cooperatively by the whole wave (8 lane groups stride over one row's sources);
//   * table slots run as one extra pseudo K block: their pre-transformed source rows are gathered
//     the same way, staged in the A image and ADDED to the accumulators in the C/D layout
//     (16 LDS reads per half instead of 32 MFMAs);
//   * HBM traffic per row and layer: one 256-B read of x, one 256-B write, ~20 B of indices
//     (neighbour re-reads hit L2: a neighborhood's rows are contiguous).
#include "common_device.hpp"

namespace desco {

constexpr int WR = 32;        // rows per wave
constexpr int NW = 8;         // waves per block (2 per SIMD: <= 256 VGPRs each)
constexpr int AH = 33;        // half-K A image row stride (floats): conflict-free ds_read_b32
constexpr int MAXS = 4;       // relation slots stored per row
constexpr int RPN = WR * MAXS + 1;
constexpr int WCAP = 128;     // source ids staged per wave (longer slices fall back to global)
constexpr int WAVE_LDS = WR * AH + 2 * RPN + 2 * WCAP;   // floats per wave

// absent sources of a batched gather step read this row instead of being predicated away
__device__ __attribute__((aligned(16))) float shmp_zero_row[64];

struct ShmpArgs {
  const float* x;
  int64_t ldx;
  const int32_t* vrowptr;
  const int32_t* vcol;
  int64_t row0, num_rows;
  int S, sm, st;
  const float* wt;
  const float* bias;
  const float* ytab;
  int64_t ldy, ytab_row0;
  float* out;
  int64_t ldo;
};

__device__ __forceinline__ void f4add(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

template <int KB, int ST>   // KB = sm + 1 resident weight blocks (1..4), ST table slots (0..2)
__global__ __launch_bounds__(NW * 64) void shmp_layer_f32_kernel(ShmpArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Bimg = lds;                                       // [KB*64][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Aw = lds + KB * 64 * 64 + wave * WAVE_LDS;        // [32][33]
  int* rpb = reinterpret_cast<int*>(Aw + WR * AH);         // 2 x [32*S+1] row pointers (absolute)
  int* ecb = rpb + 2 * RPN;                                // 2 x [WCAP] source ids

  // ---- resident weights -------------------------------------------------------------------
  for (int i = tid; i < KB * 1024; i += NW * 64)
    *reinterpret_cast<float4*>(Bimg + 4 * i) = *reinterpret_cast<const float4*>(g.wt + 4 * i);
  __syncthreads();

  // The 3 waves of a SIMD (w, w+4, w+8) run the same program; equal priority keeps them in
  // lockstep (memory phases and MFMA phases line up and add).  Distinct static priorities let one
  // wave finish its MFMA block first, so its loads/stores overlap the others' MFMAs.
  {
    const int pr = __builtin_amdgcn_readfirstlane(wave >> 2);
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr >= 3) __builtin_amdgcn_s_setprio(3);
  }
  const int g8 = lane >> 3, l8 = lane & 7;                 // 8 groups of 8 lanes: one half row each
  const int S = g.S;
  const int nslot = WR * S + 1;                            // <= 129: at most 3 per lane
  const int64_t ntiles = (g.num_rows + NW * WR - 1) / (NW * WR);

  // prologue: indices of this wave's first tile
  int64_t tile = blockIdx.x;
  {
    const int64_t w0 = tile * (NW * WR) + wave * WR;
    if (tile < ntiles && w0 < g.num_rows) {
      const int nr = (int)((g.num_rows - w0) < WR ? (g.num_rows - w0) : WR);
      const int nptr = nr * S + 1;
      for (int i = lane; i < nslot; i += 64)
        rpb[i] = g.vrowptr[(g.row0 + w0) * S + (i < nptr ? i : nptr - 1)];
      const int eb = rpb[0], ecnt = rpb[WR * S] - eb;
      for (int i = lane; i < ecnt && i < WCAP; i += 64) ecb[i] = g.vcol[eb + i];
    }
  }

  int cur = 0;
  for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    const int64_t w0 = tile * (NW * WR) + wave * WR;       // first row of this wave (relative)
    if (w0 >= g.num_rows) continue;
    const int nr = (int)((g.num_rows - w0) < WR ? (g.num_rows - w0) : WR);
    const int64_t grow0 = g.row0 + w0;
    int* rp = rpb + cur * RPN;
    int* ec = ecb + cur * WCAP;
    int* rpn = rpb + (cur ^ 1) * RPN;
    int* ecn = ecb + (cur ^ 1) * WCAP;

    // ---- prefetch the row pointers of the next tile (registers now, LDS later) --------------
    const int64_t tn = tile + gridDim.x;
    const int64_t w0n = tn * (NW * WR) + wave * WR;
    const bool has_next = tn < ntiles && w0n < g.num_rows;
    int p0 = 0, p1 = 0, p2 = 0;
    if (has_next) {
      const int nrn = (int)((g.num_rows - w0n) < WR ? (g.num_rows - w0n) : WR);
      const int nptr = nrn * S + 1;
      const int32_t* src = g.vrowptr + (g.row0 + w0n) * S;
      p0 = src[lane < nptr ? lane : nptr - 1];
      if (lane + 64 < nslot) p1 = src[lane + 64 < nptr ? lane + 64 : nptr - 1];
      if (lane + 128 < nslot) p2 = src[lane + 128 < nptr ? lane + 128 : nptr - 1];
    }
    int qn0 = 0, qn1 = 0;      // source ids of the next tile (registers until the tile ends)
    int ebn = 0, ecntn = 0;

    const int ebase = rp[0];
    const int cl = lane & 31;
    // ---- accumulator init: bias ----------------------------------------------------------------
    f32x16 acc0, acc1;
    {
      const float bv0 = g.bias ? g.bias[cl] : 0.f, bv1 = g.bias ? g.bias[32 + cl] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc0[i] = bv0;
        acc1[i] = bv1;
      }
    }
    // K blocks: b < KB-1 = relation slot b (gathered x rows), b == KB-1 = the row itself,
    // b == KB (ST > 0) = table pseudo block (gathered ytab rows, added in the C/D layout).
    // Lane group g8 serves rows it*8 + g8 (it = 0..3); per row BOTH 128-B halves are fetched
    // together (columns 4*l8 and 32 + 4*l8).  All loads of a step are unconditional -- absent
    // sources read a row of zeros -- so the 16 loads of a step are in flight
    // together (a conditional load is fenced by its own s_waitcnt), and the first step of block
    // b+1 is issued BEFORE the MFMAs of block b.
    constexpr int NB = KB + (ST > 0 ? 1 : 0);
    const float* xb = g.x + 4 * l8;
    const float* yb = ST > 0 ? g.ytab + 4 * l8 - g.ytab_row0 * g.ldy : nullptr;
    const float* zrow = shmp_zero_row + 4 * l8;
    float4 lo0, lo1, lo2, lo3, hi0, hi1, hi2, hi3;           // gathered sums of the current block
    float4 u00, u01, u10, u11, u20, u21, u30, u31;           // in flight: first source (lo, hi) of row it
    float4 w00, w01, w10, w11, w20, w21, w30, w31;           // in flight: second source
    int c0, c1, c2, c3, n0, n1, n2, n3;                      // source cursors [c, n) relative to ebase
#define DESCO_CUR(it_, slot_)                              \
  {                                                        \
    const int v_ = ((it_) * 8 + g8) * S + (slot_);         \
    c##it_ = rp[v_] - ebase;                               \
    n##it_ = rp[v_ + 1] - ebase;                           \
  }
#define DESCO_CURS(slot_) DESCO_CUR(0, slot_) DESCO_CUR(1, slot_) DESCO_CUR(2, slot_) DESCO_CUR(3, slot_)
  // two sources of row it_ (staged ids only: e < WCAP), unconditional loads
#define DESCO_ISSUE2(it_, base_, ld_)                                                 \
  {                                                                                   \
    const int m_ = n##it_ < WCAP ? n##it_ : WCAP;                                     \
    const bool k0_ = c##it_ < m_, k1_ = c##it_ + 1 < m_;                              \
    const int i0_ = ec[k0_ ? c##it_ : 0], i1_ = ec[k1_ ? c##it_ + 1 : 0];             \
    const float* p0_ = k0_ ? (base_) + (int64_t)i0_ * (ld_) : zrow;                   \
    const float* p1_ = k1_ ? (base_) + (int64_t)i1_ * (ld_) : zrow;                   \
    u##it_##0 = *reinterpret_cast<const float4*>(p0_);                                \
    u##it_##1 = *reinterpret_cast<const float4*>(p0_ + 32);                           \
    w##it_##0 = *reinterpret_cast<const float4*>(p1_);                                \
    w##it_##1 = *reinterpret_cast<const float4*>(p1_ + 32);                           \
    c##it_ += (k0_ ? 1 : 0) + (k1_ ? 1 : 0);                                          \
  }
#define DESCO_CONSUME2(it_)                                                           \
  {                                                                                   \
    f4add(lo##it_, u##it_##0);                                                        \
    f4add(hi##it_, u##it_##1);                                                        \
    f4add(lo##it_, w##it_##0);                                                        \
    f4add(hi##it_, w##it_##1);                                                        \
  }
#define DESCO_ZERO_SUMS()                                   \
  {                                                         \
    lo0 = make_float4(0.f, 0.f, 0.f, 0.f);                  \
    lo1 = lo0; lo2 = lo0; lo3 = lo0;                        \
    hi0 = lo0; hi1 = lo0; hi2 = lo0; hi3 = lo0;             \
  }
#define DESCO_ANY_STAGED()                                                                    \
  __any((c0 < (n0 < WCAP ? n0 : WCAP)) | (c1 < (n1 < WCAP ? n1 : WCAP)) |                     \
        (c2 < (n2 < WCAP ? n2 : WCAP)) | (c3 < (n3 < WCAP ? n3 : WCAP)))
  // the row itself: rows beyond nr re-read the wave's last valid row (never stored)
#define DESCO_ISSUE_SELF(it_)                                                                  \
  {                                                                                            \
    const int r_ = (it_) * 8 + g8;                                                             \
    const float* p_ = xb + (grow0 + (r_ < nr ? r_ : nr - 1)) * g.ldx;                          \
    u##it_##0 = *reinterpret_cast<const float4*>(p_);                                          \
    u##it_##1 = *reinterpret_cast<const float4*>(p_ + 32);                                     \
  }
  // table pseudo block: the first source of table slot 0 (-> u) and of table slot 1 (-> w) of row it_
#define DESCO_TAB_CUR(it_)                                                                  \
  const int v_ = ((it_) * 8 + g8) * S + g.sm;                                               \
  const int ca_ = rp[v_] - ebase, na_ = rp[v_ + 1] - ebase;                                 \
  const int nb_ = ST > 1 ? rp[v_ + 2] - ebase : na_;                                        \
  const bool k0_ = ca_ < (na_ < WCAP ? na_ : WCAP);                                         \
  const bool k1_ = ST > 1 && na_ < (nb_ < WCAP ? nb_ : WCAP);
#define DESCO_ISSUE_TAB(it_)                                                                \
  {                                                                                         \
    DESCO_TAB_CUR(it_)                                                                      \
    const int i0_ = ec[k0_ ? ca_ : 0], i1_ = ec[k1_ ? na_ : 0];                             \
    const float* p0_ = k0_ ? yb + (int64_t)i0_ * g.ldy : zrow;                              \
    u##it_##0 = *reinterpret_cast<const float4*>(p0_);                                      \
    u##it_##1 = *reinterpret_cast<const float4*>(p0_ + 32);                                 \
    if (ST > 1) {                                                                           \
      const float* p1_ = k1_ ? yb + 64 + (int64_t)i1_ * g.ldy : zrow;                       \
      w##it_##0 = *reinterpret_cast<const float4*>(p1_);                                    \
      w##it_##1 = *reinterpret_cast<const float4*>(p1_ + 32);                               \
    }                                                                                       \
  }
  // consume the step; leave the cursor of table slot 0 in (c, n) and of slot 1 in (d, m)
#define DESCO_CONSUME_TAB(it_)                                                              \
  {                                                                                         \
    DESCO_TAB_CUR(it_)                                                                      \
    f4add(lo##it_, u##it_##0);                                                              \
    f4add(hi##it_, u##it_##1);                                                              \
    if (ST > 1) {                                                                           \
      f4add(lo##it_, w##it_##0);                                                            \
      f4add(hi##it_, w##it_##1);                                                            \
    }                                                                                       \
    c##it_ = ca_ + (k0_ ? 1 : 0);                                                           \
    n##it_ = na_;                                                                           \
    d##it_ = na_ + (k1_ ? 1 : 0);                                                           \
    m##it_ = nb_;                                                                           \
  }
  // heavy rows (hub / canonical rows of dense neighborhoods, or ids beyond the staged WCAP): the
  // whole wave cooperates on one row at a time -- lane group k takes sources c+k, c+k+8, ... and
  // the 8 partial sums are folded with three xor-shuffles (lanes with equal l8 hold the same columns)
#define DESCO_COOP(it_, base_, ld_)                                                       \
  {                                                                                       \
    unsigned long long m_ = __ballot(c##it_ < n##it_);                                    \
    while (m_) {                                                                          \
      const int sl_ = __builtin_ctzll(m_);                                                \
      const int og_ = sl_ >> 3;                                                           \
      const int cc_ = __shfl(c##it_, sl_, 64), nn_ = __shfl(n##it_, sl_, 64);             \
      float4 p_ = make_float4(0.f, 0.f, 0.f, 0.f), q_ = p_;                               \
      for (int e_ = cc_ + g8; e_ < nn_; e_ += 8) {                                        \
        const int64_t j_ = e_ < WCAP ? ec[e_] : g.vcol[ebase + e_];                       \
        const float* s_ = (base_) + j_ * (ld_);                                           \
        const float4 v0_ = *reinterpret_cast<const float4*>(s_);                          \
        const float4 v1_ = *reinterpret_cast<const float4*>(s_ + 32);                     \
        f4add(p_, v0_);                                                                   \
        f4add(q_, v1_);                                                                   \
      }                                                                                   \
      _Pragma("unroll") for (int o_ = 8; o_ < 64; o_ <<= 1) {                             \
        p_.x += __shfl_xor(p_.x, o_, 64);                                                 \
        p_.y += __shfl_xor(p_.y, o_, 64);                                                 \
        p_.z += __shfl_xor(p_.z, o_, 64);                                                 \
        p_.w += __shfl_xor(p_.w, o_, 64);                                                 \
        q_.x += __shfl_xor(q_.x, o_, 64);                                                 \
        q_.y += __shfl_xor(q_.y, o_, 64);                                                 \
        q_.z += __shfl_xor(q_.z, o_, 64);                                                 \
        q_.w += __shfl_xor(q_.w, o_, 64);                                                 \
      }                                                                                   \
      if (g8 == og_) {                                                                    \
        f4add(lo##it_, p_);                                                               \
        f4add(hi##it_, q_);                                                               \
        c##it_ = n##it_;                                                                  \
      }                                                                                   \
      m_ &= ~(0xffULL << (og_ * 8));                                                      \
    }                                                                                     \
  }
  // finish a gathered block whose first step is already in flight: consume it, one more batched
  // step for rows with 3-4 sources, then the cooperative path for what is left
#define DESCO_FINISH(base_, ld_)                                                           \
  {                                                                                        \
    DESCO_CONSUME2(0) DESCO_CONSUME2(1) DESCO_CONSUME2(2) DESCO_CONSUME2(3)                \
    if (DESCO_ANY_STAGED()) {                                                              \
      DESCO_ISSUE2(0, base_, ld_) DESCO_ISSUE2(1, base_, ld_)                              \
      DESCO_ISSUE2(2, base_, ld_) DESCO_ISSUE2(3, base_, ld_)                              \
      DESCO_CONSUME2(0) DESCO_CONSUME2(1) DESCO_CONSUME2(2) DESCO_CONSUME2(3)              \
    }                                                                                      \
    if (__any((c0 < n0) | (c1 < n1) | (c2 < n2) | (c3 < n3))) {                            \
      DESCO_COOP(0, base_, ld_) DESCO_COOP(1, base_, ld_) DESCO_COOP(2, base_, ld_)        \
      DESCO_COOP(3, base_, ld_)                                                            \
    }                                                                                      \
  }
  // first step of block b_ (cursors + loads); nothing waits on the loads here
#define DESCO_ISSUE_BLOCK(b_)                                                              \
  {                                                                                        \
    if ((b_) < KB - 1) {                                                                   \
      DESCO_CURS(b_)                                                                       \
      DESCO_ISSUE2(0, xb, g.ldx) DESCO_ISSUE2(1, xb, g.ldx)                                \
      DESCO_ISSUE2(2, xb, g.ldx) DESCO_ISSUE2(3, xb, g.ldx)                                \
    } else if ((b_) == KB - 1) {                                                           \
      DESCO_ISSUE_SELF(0) DESCO_ISSUE_SELF(1) DESCO_ISSUE_SELF(2) DESCO_ISSUE_SELF(3)      \
    } else {                                                                               \
      DESCO_ISSUE_TAB(0) DESCO_ISSUE_TAB(1) DESCO_ISSUE_TAB(2) DESCO_ISSUE_TAB(3)          \
    }                                                                                      \
  }
  // write one half image (every lane writes: row = it*8 + g8, 4 floats at 4*l8)
#define DESCO_PUT(av_, it_)                             \
  {                                                     \
    float* d_ = Aw + ((it_) * 8 + g8) * AH + 4 * l8;    \
    d_[0] = av_.x;                                      \
    d_[1] = av_.y;                                      \
    d_[2] = av_.z;                                      \
    d_[3] = av_.w;                                      \
  }
  // 32 MFMAs on the staged half h_ of K block b_: A[i = lane&31][k = lane>>5], B[k][j = lane&31]
#define DESCO_MFMA_HALF(b_, h_)                                                            \
  {                                                                                        \
    const float* as_ = Aw + (lane & 31) * AH + (lane >> 5);                                \
    const float* bs_ = Bimg + ((b_) * 64 + (h_) * 32 + (lane >> 5)) * 64 + (lane & 31);    \
    _Pragma("unroll") for (int kk = 0; kk < 16; ++kk) {                                    \
      const float a_ = as_[2 * kk];                                                        \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, bs_[2 * kk * 64], acc0, 0, 0, 0);    \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, bs_[2 * kk * 64 + 32], acc1, 0, 0, 0); \
    }                                                                                      \
  }
  // add the staged table half rows in the C/D layout: acc[reg] += stage[row(reg)][lane&31]
#define DESCO_TAB_HALF(acc_)                                                               \
  {                                                                                        \
    _Pragma("unroll") for (int reg = 0; reg < 16; ++reg)                                   \
        acc_[reg] += Aw[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * AH + cl];         \
  }

    DESCO_ISSUE_BLOCK(0)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      // ---- complete the gather of block b ------------------------------------------------------
      DESCO_ZERO_SUMS()
      if (b < KB - 1) {
        DESCO_FINISH(xb, g.ldx)
      } else if (b == KB - 1) {
        lo0 = u00; hi0 = u01; lo1 = u10; hi1 = u11;
        lo2 = u20; hi2 = u21; lo3 = u30; hi3 = u31;
      } else {
        // canonical->count relations have at most one source per row: one step covers both table
        // slots; anything beyond that (general inputs) takes the cooperative path
        int d0, d1, d2, d3, m0, m1, m2, m3;
        DESCO_CONSUME_TAB(0) DESCO_CONSUME_TAB(1) DESCO_CONSUME_TAB(2) DESCO_CONSUME_TAB(3)
        if (__any((c0 < n0) | (c1 < n1) | (c2 < n2) | (c3 < n3))) {
          DESCO_COOP(0, yb, g.ldy) DESCO_COOP(1, yb, g.ldy) DESCO_COOP(2, yb, g.ldy)
          DESCO_COOP(3, yb, g.ldy)
        }
        if (ST > 1 && __any((d0 < m0) | (d1 < m1) | (d2 < m2) | (d3 < m3))) {
          c0 = d0; c1 = d1; c2 = d2; c3 = d3;
          n0 = m0; n1 = m1; n2 = m2; n3 = m3;
          DESCO_COOP(0, yb + 64, g.ldy) DESCO_COOP(1, yb + 64, g.ldy) DESCO_COOP(2, yb + 64, g.ldy)
          DESCO_COOP(3, yb + 64, g.ldy)
        }
      }
      if (b == 0 && has_next) {
        // next tile's row pointers have landed: publish them, then fetch its source ids
        rpn[lane] = p0;
        if (lane + 64 < nslot) rpn[lane + 64] = p1;
        if (lane + 128 < nslot) rpn[lane + 128] = p2;
        ebn = rpn[0];
        ecntn = rpn[WR * S] - ebn;
        if (lane < ecntn) qn0 = g.vcol[ebn + lane];
        if (lane + 64 < ecntn) qn1 = g.vcol[ebn + lane + 64];
      }
      // ---- the two 32-column halves of block b; the first gather step of block b+1 goes out
      //      under this block's MFMAs (after the low halves have left their registers)
      DESCO_PUT(lo0, 0) DESCO_PUT(lo1, 1) DESCO_PUT(lo2, 2) DESCO_PUT(lo3, 3)
      if (b + 1 < NB) DESCO_ISSUE_BLOCK(b + 1)
      if (b < KB) DESCO_MFMA_HALF(b, 0) else DESCO_TAB_HALF(acc0)
      DESCO_PUT(hi0, 0) DESCO_PUT(hi1, 1) DESCO_PUT(hi2, 2) DESCO_PUT(hi3, 3)
      if (b < KB) DESCO_MFMA_HALF(b, 1) else DESCO_TAB_HALF(acc1)
    }
#undef DESCO_CUR
#undef DESCO_CURS
#undef DESCO_ISSUE2
#undef DESCO_CONSUME2
#undef DESCO_ZERO_SUMS
#undef DESCO_ANY_STAGED
#undef DESCO_ISSUE_SELF
#undef DESCO_TAB_CUR
#undef DESCO_ISSUE_TAB
#undef DESCO_CONSUME_TAB
#undef DESCO_COOP
#undef DESCO_FINISH
#undef DESCO_ISSUE_BLOCK
#undef DESCO_PUT
#undef DESCO_MFMA_HALF
#undef DESCO_TAB_HALF

    if (has_next) {   // publish the next tile's source ids
      if (lane < ecntn) ecn[lane] = qn0;
      if (lane + 64 < ecntn && lane + 64 < WCAP) ecn[lane + 64] = qn1;
    }

    // ---- epilogue: relu + store; C/D map col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5) ----
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      if (r < nr) {
        const float v0 = acc0[reg], v1 = acc1[reg];
        float* o = g.out + (grow0 + r) * g.ldo + cl;
        o[0] = v0 > 0.f ? v0 : 0.f;
        o[32] = v1 > 0.f ? v1 : 0.f;
      }
    }
  }
}

}  // namespace desco

extern "C" int desco_shmp_layer_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                    const int32_t* vcol, int64_t row0, int64_t num_rows,
                                    int slots_stored, int slots_mfma, int slots_table,
                                    const float* wt, const float* bias, const float* ytab,
                                    int64_t ldy, int64_t ytab_row0, float* out, int64_t ldo,
                                    desco_stream_t stream) {
  using namespace desco;
  if (num_rows == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!x || !vrowptr || !wt || !out || row0 < 0 || num_rows < 0 || slots_mfma < 0 ||
      slots_mfma > 3 || slots_table < 0 || slots_mfma + slots_table > slots_stored || slots_stored < 1 ||
      slots_stored > MAXS || slots_table > 2 || (slots_table > 0 && !ytab) || ldx % 4 || mis16(x) || mis16(wt) ||
      x == out)
    return fail(DESCO_EINVAL, "desco_shmp_layer_f32: bad argument (slots_mfma <= 3, slots_table <= 2)");
  const int64_t ntiles = (num_rows + NW * WR - 1) / (NW * WR);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  const int kb = slots_mfma + 1;
  const size_t shmem = sizeof(float) * ((size_t)kb * 64 * 64 + (size_t)NW * WAVE_LDS);
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  ShmpArgs g{x,   ldx,  vrowptr, vcol, row0,      num_rows, slots_stored, slots_mfma, slots_table,
             wt,  bias, ytab,    ldy,  ytab_row0, out,      ldo};
  hipStream_t st = (hipStream_t)stream;
#define DESCO_LAUNCH(KB_, ST_)                                                                   \
  {                                                                                              \
    static bool attr_set = false;                                                                \
    if (!attr_set) {                                                                             \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shmp_layer_f32_kernel<KB_, ST_>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
      attr_set = true;                                                                           \
    }                                                                                            \
    hipLaunchKernelGGL((shmp_layer_f32_kernel<KB_, ST_>), dim3(grid), dim3(NW * 64), shmem, st,  \
                       g);                                                                       \
  }
#define DESCO_LAUNCH_ST(KB_)                    \
  switch (slots_table) {                        \
    case 0: DESCO_LAUNCH(KB_, 0) break;         \
    case 1: DESCO_LAUNCH(KB_, 1) break;         \
    default: DESCO_LAUNCH(KB_, 2) break;        \
  }
  switch (kb) {
    case 1: DESCO_LAUNCH_ST(1) break;
    case 2: DESCO_LAUNCH_ST(2) break;
    case 3: DESCO_LAUNCH_ST(3) break;
    default: DESCO_LAUNCH_ST(4) break;
  }
#undef DESCO_LAUNCH_ST
#undef DESCO_LAUNCH
  return launch_status("desco_shmp_layer_f32");
}
