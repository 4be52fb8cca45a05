// Neighborhood-resident multi-layer SHMP kernel: a workgroup carries a PACK of canonical
// neighborhoods through ALL SHMP layers with the node features resident in LDS.
//
// Reference semantics: BaseGNNCore.forward, SAGE branch, after to_hetero (gnn_model.py:230-277, 372-404)
// for the "count" and "canonical" node types -- per layer and destination type
//   x' = relu( sum_s agg_s (U_n W_s)^T + x U_x^T + bias ),   agg_s[i] = sum_{j ->_s i} x[j]
// in the folded form of DESIGN.md 4.1 (slots 0, 1: count sources by MFMA; slots 2, 3: the neighborhood's
// own canonical node, applied from the table T = x_canon [W_2 | W_3]); the first layer is the closed
// degree-affine form of the constant input (desco_degree_affine_f32); the kernel leaves, per
// neighborhood and layer, the global_add_pool sum of the count rows (gnn_model.py:88-89, 107) and the
// canonical row (operand of anchor_mlp, gnn_model.py:69-73).
//
// Why.  Canonical neighborhoods are closed systems: no edge leaves one.  The layer-by-layer kernels
// (shmp_layer16.hip) write X_l to HBM and gather it back through L2 with a dependent round trip per
// two-source step (Syn_1827 shapes: 0.30 of the HBM roofline, waves parked on memory 57 % of the time).
// Here X never leaves the CU: compulsory HBM traffic is the CSR once plus 2 x 8 x 256 B per
// neighborhood, and a gather step is an LDS read.
//
// Shape.  8 waves (2 per SIMD, 256 VGPRs), one pack at a time per workgroup, packs handed out from a
// global counter.  A pack = up to 16 consecutive neighborhoods, each starting at a 16-row tile boundary
// of the LDS image: rows 0..15 are the canonical rows of the pack's neighborhoods, then up to 31 count
// tiles.  LDS (163 360 B): X [512 rows][64] fp32 (16-byte chunks XOR-swizzled by the row's low 4 bits),
// a zero row, the table T [16][128], per-tile pooling partials [31][64], the pack's CSR as 16-bit local
// row ids + 16-bit row pointers.  No weights in LDS: the bf16 planes (6-product split, fp32-accurate)
// stream from L2 as ready-made MFMA B fragments ("fragment-major", 12 KB per 64x32 K-step), each
// fragment set shared by the wave's (up to 4) tiles and prefetched one step ahead in registers.
//
// A wave tile = 16 rows on v_mfma_f32_16x16x32_bf16.  The gather runs IN the MFMA A layout: lane
// (r = lane & 15, q = lane >> 4) sums columns 32 h + 8 q .. + 7 of the sources of row r, 32 bytes per
// source from LDS -- no staging image, one bf16 split per K-half.  Rows with more than HEAVY sources in
// a slot are summed by the whole wave (16 sources per step, row_ror reduction): the switch depends on
// the row's OWN degree only, so a row's summation order -- and with neighborhood-aligned tiles every
// pooled sum -- does not depend on where the neighborhood sits in a launch or shard.
#include <vector>

#include "common_device.hpp"

namespace desco {
namespace res {

constexpr int NWAVE = 8;
constexpr int NT = 4;                      // count tiles per wave
constexpr int XROWS = 512;                 // 16 canonical rows + 30 count tiles (+ one spare tile)
constexpr int MAXNB = 16;
constexpr int MAXCT = 30;
constexpr int ECAP = 5632;                 // directed edges (source ids) per pack
constexpr int HEAVY = 24;
constexpr int STEPS = 16;                  // fragment steps per layer: 4 table, 6 canonical, 6 count
constexpr int STEP_BYTES = 3 * 4 * 64 * 16;

constexpr int OFF_Z = XROWS * 256;                     // zero row = LDS row 512
constexpr int OFF_T = OFF_Z + 256;                     // T [16][128] f32
constexpr int OFF_P = OFF_T + MAXNB * 128 * 4;         // tile partial sums [31][64] f32
constexpr int OFF_IDS = OFF_P + MAXCT * 64 * 4;        // source ids (LDS row numbers), u16
constexpr int OFF_RP = OFF_IDS + ECAP * 2 + 16;        // row pointers [512*4 + 1] u16
constexpr int OFF_INFO = OFF_RP + (XROWS * 4 + 8) * 2;
#ifdef RES_PROF
constexpr int LDS_BYTES = OFF_INFO + 672 + 8 * 10 * 4;
#else
constexpr int LDS_BYTES = OFF_INFO + 672;
#endif
static_assert(LDS_BYTES <= 160 * 1024, "resident SHMP kernel: LDS budget exceeded");

struct Info {                 // block-shared bookkeeping of the current pack
  int nb[MAXNB];              // neighborhood index of slot j (-1: empty)
  int cs[MAXNB];              // first global count row of neighborhood j
  int n[MAXNB];               // its count rows
  int trow[MAXNB];            // its first LDS row (16 + 16 * tstart)
  int tstart[MAXNB];          // its first count tile
  int seg_src[2 * MAXNB];     // id segments: canonical row j (j < 16), count rows of neighborhood j (16 + j):
  int seg_len[2 * MAXNB];     // first global edge and length
  unsigned char tile2nb[32];
  int scan[NWAVE];
  int pack, ntiles;
};
static_assert(sizeof(Info) <= 672, "Info does not fit its LDS slot");

struct Args {
  const int32_t* count_ptr;
  const int32_t* vrowptr;
  const int32_t* vcol;
  int64_t num_count;
  const int32_t* pack_list;   // [num_packs][16] neighborhood indices, -1 = unused slot
  int num_packs;
  const float* l0coef;        // [2 types][5][64]: count (slots 0..3, self), canonical (slots 0, 1, -, -, self)
  const char* wfrag;          // [num_layers][16 steps][3 planes][4 col tiles][64 lanes][8 bf16]
  const float* bias;          // [num_layers][2 types][64]
  int num_layers;
  float* pooled;              // [B][ldp]: block (l) = sum of the count rows of x^l, l = 1 .. num_layers + 1
  int64_t ldp;
  float* canon;               // [B][ldc]: block (l) = canonical row of x^l
  int64_t ldc;
  int* counter;
};

using bf16x8 = __attribute__((ext_vector_type(8))) short;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void f4add(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

template <int CTRL>
__device__ __forceinline__ float ror_add(const float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (lane & 15), result in every lane; fixed order
__device__ __forceinline__ float row16_sum(float v) {
  v = ror_add<0x128>(v);
  v = ror_add<0x124>(v);
  v = ror_add<0x122>(v);
  v = ror_add<0x121>(v);
  return v;
}
__device__ __forceinline__ void row16_sum4(float4& v) {
  v.x = row16_sum(v.x);
  v.y = row16_sum(v.y);
  v.z = row16_sum(v.z);
  v.w = row16_sum(v.w);
}

#define RES_LDS4(off_) (*reinterpret_cast<const float4*>(lds + (off_)))
// A source id is kept as w = (LDS row << 4) | (LDS row & 15): the byte offset of the row's logical 16-byte
// chunk kap (even: the lane reads kap and kap + 1; chunks are XOR-swizzled by the row's low 4 bits) is then
// (w ^ kap) << 4 -- two instructions per source in the gather loop.
#define RES_W(row_) ((((row_)) << 4) | ((row_) & 15))
#define RES_XOFF(w_, kap_) (((w_) ^ (kap_)) << 4)
constexpr int WZERO = XROWS << 4;          // w of the zero row

// The gather of one (tile, slot, K half), software-pipelined over a wave's tiles (A layout: lane (r, q) sums
// columns 4 kap .. 4 kap + 7 of the sources of the tile's row r).  RES_G issues the loads of the row's first
// four sources (ids cached in registers per pack: hd[t]; lanes whose row has fewer read the zero row) and
// leaves them in flight -- under the previous tile's bf16 splits and MFMAs; RES_C adds them up in CSR order,
// fetches further sources four at a time, and hands rows with more than HEAVY sources to the whole wave
// (lane r' takes sources r', r' + 16, ...; row_ror reduction).  The self block (B_ == 2) is the same
// machinery with one source: the row itself.
#define RES_G(T_, TROW_, B_, KAP_)                                                                \
  {                                                                                               \
    unsigned w0_, w1_;                                                                            \
    if ((B_) == 2) {                                                                              \
      g_ch = 0;                                                                                   \
      g_n = 1;                                                                                    \
      w0_ = (unsigned)RES_W((TROW_) + r);                                                         \
      w1_ = 0;                                                                                    \
    } else {                                                                                      \
      const int c1_ = hd[T_][0] >> 16;                                                            \
      g_ch = (B_) ? c1_ : (int)(hd[T_][0] & 0xffff);                                              \
      g_n = (B_) ? (int)hd[T_][1] : c1_;                                                          \
      w0_ = (B_) ? hd[T_][4] : hd[T_][2];                                                         \
      w1_ = (B_) ? hd[T_][5] : hd[T_][3];                                                         \
    }                                                                                             \
    g_heavy = (g_n - g_ch) > HEAVY;                                                               \
    g_rem = g_heavy ? 0 : g_n - g_ch;                                                             \
    const int j0_ = g_rem > 0 ? (int)(w0_ & 0xffff) : WZERO, j1_ = g_rem > 1 ? (int)(w0_ >> 16) : WZERO; \
    const int j2_ = g_rem > 2 ? (int)(w1_ & 0xffff) : WZERO, j3_ = g_rem > 3 ? (int)(w1_ >> 16) : WZERO; \
    const int a0_ = RES_XOFF(j0_, KAP_), a1_ = RES_XOFF(j1_, KAP_);                               \
    const int a2_ = RES_XOFF(j2_, KAP_), a3_ = RES_XOFF(j3_, KAP_);                               \
    gv0 = RES_LDS4(a0_);                                                                          \
    gv1 = RES_LDS4(a0_ ^ 16);                                                                     \
    gv2 = RES_LDS4(a1_);                                                                          \
    gv3 = RES_LDS4(a1_ ^ 16);                                                                     \
    gv4 = RES_LDS4(a2_);                                                                          \
    gv5 = RES_LDS4(a2_ ^ 16);                                                                     \
    gv6 = RES_LDS4(a3_);                                                                          \
    gv7 = RES_LDS4(a3_ ^ 16);                                                                     \
    g_rem -= 4;                                                                                   \
  }
#define RES_C(KAP_, S0_, S1_)                                                                     \
  {                                                                                               \
    S0_ = gv0;                                                                                    \
    S1_ = gv1;                                                                                    \
    f4add(S0_, gv2);                                                                              \
    f4add(S1_, gv3);                                                                              \
    f4add(S0_, gv4);                                                                              \
    f4add(S1_, gv5);                                                                              \
    f4add(S0_, gv6);                                                                              \
    f4add(S1_, gv7);                                                                              \
    if (__any(g_rem > 0)) {                                                                       \
      const unsigned short* ip_ = ids + g_ch + 4;                                                 \
      int rem_ = g_rem;                                                                           \
      do {                                                                                        \
        const int i0_ = ip_[0], i1_ = ip_[1], i2_ = ip_[2], i3_ = ip_[3];                         \
        const int j0_ = rem_ > 0 ? i0_ : WZERO, j1_ = rem_ > 1 ? i1_ : WZERO;                     \
        const int j2_ = rem_ > 2 ? i2_ : WZERO, j3_ = rem_ > 3 ? i3_ : WZERO;                     \
        const int a0_ = RES_XOFF(j0_, KAP_), a1_ = RES_XOFF(j1_, KAP_);                           \
        const int a2_ = RES_XOFF(j2_, KAP_), a3_ = RES_XOFF(j3_, KAP_);                           \
        const float4 v00_ = RES_LDS4(a0_), v01_ = RES_LDS4(a0_ ^ 16);                             \
        const float4 v10_ = RES_LDS4(a1_), v11_ = RES_LDS4(a1_ ^ 16);                             \
        const float4 v20_ = RES_LDS4(a2_), v21_ = RES_LDS4(a2_ ^ 16);                             \
        const float4 v30_ = RES_LDS4(a3_), v31_ = RES_LDS4(a3_ ^ 16);                             \
        ip_ += 4;                                                                                 \
        rem_ -= 4;                                                                                \
        f4add(S0_, v00_);                                                                         \
        f4add(S1_, v01_);                                                                         \
        f4add(S0_, v10_);                                                                         \
        f4add(S1_, v11_);                                                                         \
        f4add(S0_, v20_);                                                                         \
        f4add(S1_, v21_);                                                                         \
        f4add(S0_, v30_);                                                                         \
        f4add(S1_, v31_);                                                                         \
      } while (__any(rem_ > 0));                                                                  \
    }                                                                                             \
    unsigned long long hm_ = __ballot(g_heavy) & 0xffffULL;                                       \
    while (hm_) {                                                                                 \
      const int R_ = __builtin_ctzll(hm_);                                                        \
      hm_ &= hm_ - 1;                                                                             \
      const int cc_ = __shfl(g_ch, R_, 64), nn_ = __shfl(g_n, R_, 64);                            \
      float4 p0_ = make_float4(0.f, 0.f, 0.f, 0.f), p1_ = p0_;                                    \
      for (int e_ = cc_ + r; e_ < nn_; e_ += 16) {                                                \
        const int a_ = RES_XOFF((int)ids[e_], KAP_);                                              \
        const float4 v0_ = RES_LDS4(a_), v1_ = RES_LDS4(a_ ^ 16);                                 \
        f4add(p0_, v0_);                                                                          \
        f4add(p1_, v1_);                                                                          \
      }                                                                                           \
      row16_sum4(p0_);                                                                            \
      row16_sum4(p1_);                                                                            \
      if (r == R_) {                                                                              \
        S0_ = p0_;                                                                                \
        S1_ = p1_;                                                                                \
      }                                                                                           \
    }                                                                                             \
  }
// 8 floats (k = 8 q .. 8 q + 7 of a K half) -> the three bf16 A fragments
#define RES_SPLIT(S0_, S1_, AH_, AM_, AL_)                                   \
  {                                                                          \
    uint32_t h0_, h1_, h2_, h3_, m0_, m1_, m2_, m3_, l0_, l1_, l2_, l3_;     \
    split2_bf16x3(S0_.x, S0_.y, h0_, m0_, l0_);                              \
    split2_bf16x3(S0_.z, S0_.w, h1_, m1_, l1_);                              \
    split2_bf16x3(S1_.x, S1_.y, h2_, m2_, l2_);                              \
    split2_bf16x3(S1_.z, S1_.w, h3_, m3_, l3_);                              \
    AH_ = __builtin_bit_cast(bf16x8, (u32x4){h0_, h1_, h2_, h3_});           \
    AM_ = __builtin_bit_cast(bf16x8, (u32x4){m0_, m1_, m2_, m3_});           \
    AL_ = __builtin_bit_cast(bf16x8, (u32x4){l0_, l1_, l2_, l3_});           \
  }
#if defined(RES_ABL) && RES_ABL == 2          // ablation (A/B builds only): no matrix work
#define RES_M16(a_, b_, c_) c_[0] += (float)(a_[0] + b_[0]);
#else
#define RES_M16(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0);
#endif
// the six products of weight >= 2^-16, smallest first, on the four 16-column tiles; BS_[p * 4 + t]
#define RES_MFMA24(AH_, AM_, AL_, BS_, Q_)                                                    \
  {                                                                                           \
    RES_M16(AL_, BS_[0], Q_[0]) RES_M16(AL_, BS_[1], Q_[1])                                   \
    RES_M16(AH_, BS_[8], Q_[0]) RES_M16(AH_, BS_[9], Q_[1])                                   \
    RES_M16(AM_, BS_[4], Q_[0]) RES_M16(AM_, BS_[5], Q_[1])                                   \
    RES_M16(AM_, BS_[0], Q_[0]) RES_M16(AM_, BS_[1], Q_[1])                                   \
    RES_M16(AH_, BS_[4], Q_[0]) RES_M16(AH_, BS_[5], Q_[1])                                   \
    RES_M16(AH_, BS_[0], Q_[0]) RES_M16(AH_, BS_[1], Q_[1])                                   \
    RES_M16(AL_, BS_[2], Q_[2]) RES_M16(AL_, BS_[3], Q_[3])                                   \
    RES_M16(AH_, BS_[10], Q_[2]) RES_M16(AH_, BS_[11], Q_[3])                                 \
    RES_M16(AM_, BS_[6], Q_[2]) RES_M16(AM_, BS_[7], Q_[3])                                   \
    RES_M16(AM_, BS_[2], Q_[2]) RES_M16(AM_, BS_[3], Q_[3])                                   \
    RES_M16(AH_, BS_[6], Q_[2]) RES_M16(AH_, BS_[7], Q_[3])                                   \
    RES_M16(AH_, BS_[2], Q_[2]) RES_M16(AH_, BS_[3], Q_[3])                                   \
  }
// one fragment set: 12 x 16 bytes per lane, contiguous 1 KB per wave instruction
#define RES_LOADB(BS_, PTR_)                                                                   \
  {                                                                                            \
    const char* p_ = (PTR_) + lane * 16;                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_)                                          \
        BS_[i_] = *reinterpret_cast<const bf16x8*>(p_ + i_ * 1024);                            \
  }
// K step (slot B_ of K half H_) over the wave's count tiles, weights BS_
#define RES_STEP(B_, H_, BS_)                                                                 \
  {                                                                                           \
    _Pragma("unroll") for (int t_ = 0; t_ < NT; ++t_) {                                       \
      if (t_ < ntl) RES_STEP1(t_, RES_TROW(t_), B_, H_, BS_)                                  \
    }                                                                                         \
  }
// one tile: gather (or the row itself), bf16 split, 24 MFMAs
#define RES_STEP1(T_, TROW_, B_, H_, BS_)                                                     \
  {                                                                                           \
    const int kap_ = 8 * (H_) + 2 * q;                                                        \
    if ((B_) == 2 || ((live >> (2 * (T_) + (B_))) & 1)) {                                     \
      float4 s0_, s1_;                                                                        \
      bf16x8 ah_, am_, al_;                                                                   \
      RES_G(T_, TROW_, B_, kap_)                                                              \
      RES_C(kap_, s0_, s1_)                                                                   \
      RES_SPLIT(s0_, s1_, ah_, am_, al_)                                                      \
      RES_MFMA24(ah_, am_, al_, BS_, acc[T_])                                                 \
    }                                                                                         \
  }
// first LDS row of the tile in slot t_ of this wave: count tile k = 8 t + pw (t < 3), 24 + wave - 1 (t = 3, waves
// 1..6); slot 3 of wave 7 = the canonical tile (rows 0..15); wave 0 keeps slot 3 free (it computes the table)
#define RES_TK(t_) ((t_) < 3 ? 8 * (t_) + pw : 23 + wave)
#define RES_TROW(t_) ((wave == 7 && (t_) == 3) ? 0 : 16 + 16 * RES_TK(t_))

#ifdef RES_PROF      // cycle-counter build (tools/debug/ab_resident.sh prof): s_memtime per phase, wave 0 / other waves
__device__ unsigned long long res_prof[32];
#define RES_T(ph_)                                                                            \
  {                                                                                           \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                             \
    if (lane == 0) profl[wave * 10 + (ph_)] += (unsigned)(now_ - tprof);                       \
    tprof = now_;                                                                             \
  }
#else
#define RES_T(ph_)
#endif

__global__ __launch_bounds__(NWAVE * 64) void shmp_resident_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  unsigned short* ids = reinterpret_cast<unsigned short*>(lds + OFF_IDS);
  unsigned short* rp = reinterpret_cast<unsigned short*>(lds + OFF_RP);
  float* Tl = reinterpret_cast<float*>(lds + OFF_T);
  float* Pl = reinterpret_cast<float*>(lds + OFF_P);
  Info* info = reinterpret_cast<Info*>(lds + OFF_INFO);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;      // A / B operand lane map; C / D: column c = r, row group g = q
  const int64_t Nc = a.num_count;

  if (tid < 64) reinterpret_cast<float*>(lds + OFF_Z)[tid] = 0.f;
  if (tid < 8) ids[ECAP + tid] = 0;
#ifdef RES_PROF
  unsigned* profl = reinterpret_cast<unsigned*>(lds + OFF_INFO + 672);
  if (tid < NWAVE * 10) profl[tid] = 0;
  unsigned long long tprof = __builtin_amdgcn_s_memtime();
#endif

  for (;;) {
    // ================= next pack ==================================================================
    if (tid == 0) info->pack = atomicAdd(a.counter, 1);
    __syncthreads();
    const int pack = info->pack;
    if (pack >= a.num_packs) break;
    if (tid < 64) {
      const int j = lane;
      const int nbj = j < MAXNB ? a.pack_list[(int64_t)pack * MAXNB + j] : -1;
      int cs = 0, n = 0;
      if (nbj >= 0) {
        cs = a.count_ptr[nbj];
        n = a.count_ptr[nbj + 1] - cs;
      }
      const int tiles = (n + 15) >> 4;
      int incl = tiles;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      const int ts = incl - tiles;
      if (j < MAXNB) {
        info->nb[j] = nbj;
        info->cs[j] = cs;
        info->n[j] = n;
        info->tstart[j] = ts;
        info->trow[j] = 16 + 16 * ts;
        for (int k = ts; k < ts + tiles; ++k) info->tile2nb[k] = (unsigned char)j;
        int s0 = 0, l0 = 0, s1 = 0, l1 = 0;
        if (nbj >= 0) {
          const int32_t* vk = a.vrowptr + (Nc + nbj) * 4;
          s0 = vk[0];
          l0 = vk[4] - s0;
          s1 = a.vrowptr[(int64_t)cs * 4];
          l1 = a.vrowptr[(int64_t)(cs + n) * 4] - s1;
        }
        info->seg_src[j] = s0;
        info->seg_len[j] = l0;
        info->seg_src[MAXNB + j] = s1;
        info->seg_len[MAXNB + j] = l1;
      }
      if (j == MAXNB - 1) info->ntiles = incl;
      const unsigned long long vm = __ballot(nbj >= 0);
      if (j == 0) info->scan[0] = __builtin_popcountll(vm);
    }
    __syncthreads();
    const int ntiles = info->ntiles;
    const int nnb = info->scan[0];
    __syncthreads();          // (scan[] is reused by the row-pointer scan below)

    // ---- CSR of the pack: one thread per LDS row ---------------------------------------------------
    {
      const int rho = tid;
      int j = 0, o = 0;
      bool valid;
      int64_t grow = 0;
      if (rho < 16) {
        j = rho;
        valid = j < nnb;
        if (valid) grow = Nc + info->nb[j];
      } else {
        const int k = (rho - 16) >> 4;
        valid = k < ntiles;
        if (valid) {
          j = info->tile2nb[k];
          o = rho - info->trow[j];
          valid = o < info->n[j];
          grow = (int64_t)info->cs[j] + o;
        }
      }
      int v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
      if (valid) {
        const int32_t* vp = a.vrowptr + grow * 4;
        v0 = vp[0];
        v1 = vp[1];
        v2 = vp[2];
        v3 = vp[3];
        v4 = vp[4];
      }
      const int tot = v4 - v0;
      int incl = tot;
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) {
        const int t = __shfl_up(incl, s, 64);
        if (lane >= s) incl += t;
      }
      if (lane == 63) info->scan[wave] = incl;
      __syncthreads();
      int base = incl - tot;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w)
        if (w < wave) base += info->scan[w];
      rp[4 * rho + 0] = (unsigned short)base;
      rp[4 * rho + 1] = (unsigned short)(base + (v1 - v0));
      rp[4 * rho + 2] = (unsigned short)(base + (v2 - v0));
      rp[4 * rho + 3] = (unsigned short)(base + (v3 - v0));
      if (rho == XROWS - 1) rp[4 * XROWS] = (unsigned short)(base + tot);
    }
    __syncthreads();
    // ---- source ids: the edges of a canonical row / of a neighborhood's count rows are one contiguous run
    //      of vcol and of the LDS id array; the whole block copies each run (coalesced), global id -> LDS row
    {
      const int cbase = (int)Nc;
      for (int sgm = 0; sgm < 2 * MAXNB; ++sgm) {
        const int j = sgm & (MAXNB - 1);
        if (j >= nnb) continue;
        const int len = info->seg_len[sgm];
        const int32_t* src = a.vcol + info->seg_src[sgm];
        const int trj = info->trow[j], csj = info->cs[j];
        unsigned short* dst = ids + rp[4 * (sgm < MAXNB ? j : trj)];
        for (int e = tid; e < len; e += NWAVE * 64) {
          const int gid = src[e];
          const int row = gid >= cbase ? j : trj + (gid - csj);      // its own canonical node, or a count row
          dst[e] = (unsigned short)RES_W(row);
        }
      }
    }
    __syncthreads();

    RES_T(0)
    // ---- this wave's tiles (see RES_TK) and their CSR heads, cached in registers for all layers --------
    const int pw = wave == 0 ? 6 : (wave == 7 ? 7 : wave - 1);
    unsigned hd[NT][6];
    unsigned live = 0;             // bit 2 t + s: slot s of tile t has a source; bit 8 + t: a canonical source
    unsigned tnr4 = 0;             // valid rows of tile t in bits 5 t .. 5 t + 4
    int ntl = 0;                   // count tile slots in use
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool special = t == NT - 1 && (wave == 0 || wave == 7);
      const int k = RES_TK(t);
      const bool ok = !special && k < ntiles;
      const bool canon = special && wave == 7;
#pragma unroll
      for (int i = 0; i < 6; ++i) hd[t][i] = 0;
      if (ok || canon) {
        const unsigned short* rr = rp + (RES_TROW(t) + r) * 4;
        const int c0 = rr[0], c1 = rr[1], c2 = rr[2], c4 = rr[4];
        hd[t][0] = (unsigned)c0 | ((unsigned)c1 << 16);
        hd[t][1] = (unsigned)c2;
        const unsigned short* ia = ids + c0;
        const unsigned short* ib = ids + c1;
        hd[t][2] = (unsigned)ia[0] | ((unsigned)ia[1] << 16);
        hd[t][3] = (unsigned)ia[2] | ((unsigned)ia[3] << 16);
        hd[t][4] = (unsigned)ib[0] | ((unsigned)ib[1] << 16);
        hd[t][5] = (unsigned)ib[2] | ((unsigned)ib[3] << 16);
        if (__any(c1 > c0)) live |= 1u << (2 * t);
        if (__any(c2 > c1)) live |= 2u << (2 * t);
        if (ok) {
          ntl = t + 1;
          if (__any(c4 > c2)) live |= 0x100u << t;
          const int j = info->tile2nb[k];
          const int left = info->n[j] - 16 * (k - info->tstart[j]);
          tnr4 |= (unsigned)(left < 16 ? left : 16) << (5 * t);
        }
      }
    }
    live = __builtin_amdgcn_readfirstlane(live);
    tnr4 = __builtin_amdgcn_readfirstlane(tnr4);
    ntl = __builtin_amdgcn_readfirstlane(ntl);
#if defined(RES_ABL) && RES_ABL == 1          // ablation (A/B builds only): no gathers
    live = 0;
#endif

    // ---- first layer, closed form (constant input): x1 = relu(sum_s deg_s coef_s + coef_self) ------
    // four rows per step: lane (rr = lane >> 4, cq = lane & 15) owns columns 4 cq .. 4 cq + 3 of row 4 i + rr
    {
      const int rr = lane >> 4, cq = lane & 15;
      float4 cc[5], ck[3];
#pragma unroll
      for (int s_ = 0; s_ < 5; ++s_) cc[s_] = *reinterpret_cast<const float4*>(a.l0coef + s_ * 64 + 4 * cq);
      ck[0] = *reinterpret_cast<const float4*>(a.l0coef + 5 * 64 + 4 * cq);
      ck[1] = *reinterpret_cast<const float4*>(a.l0coef + 6 * 64 + 4 * cq);
      ck[2] = *reinterpret_cast<const float4*>(a.l0coef + 9 * 64 + 4 * cq);
      float4 run = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int rho = wave * 64 + 4 * i + rr;
        const unsigned short* rq = rp + rho * 4;
        const int e0 = rq[0], e1 = rq[1], e2 = rq[2], e3 = rq[3], e4 = rq[4];
        const float d0 = (float)(e1 - e0), d1 = (float)(e2 - e1), d2 = (float)(e3 - e2), d3 = (float)(e4 - e3);
        const bool canon_row = wave == 0 && i < 4;                    // rows 0..15
        bool valid;
        float4 v;
        if (canon_row) {
          valid = rho < nnb;
          v.x = fmaf(d1, ck[1].x, fmaf(d0, ck[0].x, ck[2].x));
          v.y = fmaf(d1, ck[1].y, fmaf(d0, ck[0].y, ck[2].y));
          v.z = fmaf(d1, ck[1].z, fmaf(d0, ck[0].z, ck[2].z));
          v.w = fmaf(d1, ck[1].w, fmaf(d0, ck[0].w, ck[2].w));
        } else {
          const int k = (rho - 16) >> 4;
          valid = k < ntiles;
          if (valid) {
            const int j = info->tile2nb[k];
            valid = rho - info->trow[j] < info->n[j];
          }
          v.x = fmaf(d3, cc[3].x, fmaf(d2, cc[2].x, fmaf(d1, cc[1].x, fmaf(d0, cc[0].x, cc[4].x))));
          v.y = fmaf(d3, cc[3].y, fmaf(d2, cc[2].y, fmaf(d1, cc[1].y, fmaf(d0, cc[0].y, cc[4].y))));
          v.z = fmaf(d3, cc[3].z, fmaf(d2, cc[2].z, fmaf(d1, cc[1].z, fmaf(d0, cc[0].z, cc[4].z))));
          v.w = fmaf(d3, cc[3].w, fmaf(d2, cc[2].w, fmaf(d1, cc[1].w, fmaf(d0, cc[0].w, cc[4].w))));
        }
        v.x = valid ? fmaxf(v.x, 0.f) : 0.f;
        v.y = valid ? fmaxf(v.y, 0.f) : 0.f;
        v.z = valid ? fmaxf(v.z, 0.f) : 0.f;
        v.w = valid ? fmaxf(v.w, 0.f) : 0.f;
        *reinterpret_cast<float4*>(lds + (rho << 8) + (((cq ^ rho) & 15) << 4)) = v;
        if (canon_row) {
          if (valid) *reinterpret_cast<float4*>(a.canon + (int64_t)info->nb[rho & 15] * a.ldc + 64 + 4 * cq) = v;
        } else {
          f4add(run, v);
        }
        if ((i & 3) == 3) {          // a 16-row tile is complete: sum its four row groups (lane quarters)
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          float tv[4] = {run.x, run.y, run.z, run.w};
#pragma unroll
          for (int c_ = 0; c_ < 4; ++c_) {
            u32x2 w_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(tv[c_]), __float_as_uint(tv[c_]), false, false);
            tv[c_] = __uint_as_float(w_[0]) + __uint_as_float(w_[1]);
            w_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(tv[c_]), __float_as_uint(tv[c_]), false, false);
            tv[c_] = __uint_as_float(w_[0]) + __uint_as_float(w_[1]);
          }
          const int k = wave * 4 + (i >> 2) - 1;
          if (rr == 0 && k >= 0 && k < MAXCT)
            *reinterpret_cast<float4*>(Pl + k * 64 + 4 * cq) = make_float4(tv[0], tv[1], tv[2], tv[3]);
          run = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    RES_T(1)
    __syncthreads();
    // pooled sums of x1 (and, below, of every layer): wave w adds the tile partials of neighborhoods w, w + 8
#define RES_POOL_OUT(BLK_)                                                          \
  for (int j_ = wave; j_ < nnb; j_ += NWAVE) {                                     \
    const int t0_ = info->tstart[j_], t1_ = t0_ + ((info->n[j_] + 15) >> 4);       \
    float s_ = 0.f;                                                                \
    for (int k_ = t0_; k_ < t1_; ++k_) s_ += Pl[k_ * 64 + lane];                   \
    a.pooled[(int64_t)info->nb[j_] * a.ldp + (BLK_) * 64 + lane] = s_;             \
  }
    RES_POOL_OUT(1)
    RES_T(9)

    // ================= GEMM layers ===================================================================
    // One fragment set per K step (48 registers), loaded at the top of the step: the first tile's gather runs
    // under the loads, so no second set is kept one step ahead.
    const char* wl = a.wfrag;
    for (int L = 0; L < a.num_layers; ++L, wl += STEPS * STEP_BYTES) {
      const bool last = L == a.num_layers - 1;
      f32x4 acc[NT][4];
      bf16x8 BS[12];
      float4 gv0, gv1, gv2, gv3, gv4, gv5, gv6, gv7;      // gather loads in flight (RES_G -> RES_C)
      int g_ch, g_n, g_rem;
      bool g_heavy;
      if (wave == 0) {
        // ---- table T = x_canon [W_2 | W_3] (K = 64, N = 128): steps (h0,j0) (h0,j1) (h1,j0) (h1,j1)
        bf16x8 BT[12];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
          float4 s0, s1;
          bf16x8 ah, am, al;
          RES_LOADB(BS, wl)
          RES_LOADB(BT, wl + 1 * STEP_BYTES)
          s0 = RES_LDS4(RES_XOFF(RES_W(r), 2 * q));
          s1 = RES_LDS4(RES_XOFF(RES_W(r), 2 * q) ^ 16);
          RES_SPLIT(s0, s1, ah, am, al)
          RES_MFMA24(ah, am, al, BS, acc[0])
          RES_LOADB(BS, wl + 2 * STEP_BYTES)
          RES_MFMA24(ah, am, al, BT, acc[1])
          RES_LOADB(BT, wl + 3 * STEP_BYTES)
          s0 = RES_LDS4(RES_XOFF(RES_W(r), 8 + 2 * q));
          s1 = RES_LDS4(RES_XOFF(RES_W(r), 8 + 2 * q) ^ 16);
          RES_SPLIT(s0, s1, ah, am, al)
          RES_MFMA24(ah, am, al, BS, acc[0])
          RES_MFMA24(ah, am, al, BT, acc[1])
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) Tl[(4 * q + e) * 128 + 64 * jj + 16 * u + r] = acc[jj][u][e];
        RES_T(2)
      }
      RES_T(3)
      // ---- count tiles, and the canonical destination rows (wave 7, tile slot 3 = LDS rows 0..15) --------
      {
        const float* bc = a.bias + (L * 2) * 64;
        const float b0 = bc[r], b1 = bc[16 + r], b2 = bc[32 + r], b3 = bc[48 + r];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t][0] = f32x4{b0, b0, b0, b0};
          acc[t][1] = f32x4{b1, b1, b1, b1};
          acc[t][2] = f32x4{b2, b2, b2, b2};
          acc[t][3] = f32x4{b3, b3, b3, b3};
        }
        if (wave == 7) {
          const float* bk = bc + 64;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float bv = bk[16 * u + r];
            acc[3][u] = f32x4{bv, bv, bv, bv};
          }
        }
      }
      for (int sb = 0; sb < 6; ++sb) {
        const int b = sb >> 1, h = sb & 1;
        if (wave == 7) {
          RES_LOADB(BS, wl + (4 + sb) * STEP_BYTES)
          RES_STEP1(3, 0, b, h, BS)
        }
        RES_LOADB(BS, wl + (10 + sb) * STEP_BYTES)
        RES_STEP(b, h, BS)
      }
      RES_T(4)
      __syncthreads();      // every gather of this layer is done (X may be overwritten); T is complete
      RES_T(5)

      // ---- epilogue: table relations, relu, new rows into X, pooling partial per tile ----------------
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t >= ntl) continue;
        const int trow = RES_TROW(t);
        if ((live >> (8 + t)) & 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned short* rr = rp + (trow + 4 * q + e) * 4;
            const int c2 = rr[2], c3 = rr[3], c4 = rr[4];
            if (c3 > c2) {
              const float* tp = Tl + ((int)ids[c2] >> 4) * 128 + r;
              acc[t][0][e] += tp[0];
              acc[t][1][e] += tp[16];
              acc[t][2][e] += tp[32];
              acc[t][3][e] += tp[48];
            }
            if (c4 > c3) {
              const float* tp = Tl + ((int)ids[c3] >> 4) * 128 + 64 + r;
              acc[t][0][e] += tp[0];
              acc[t][1][e] += tp[16];
              acc[t][2][e] += tp[32];
              acc[t][3][e] += tp[48];
            }
          }
        }
        const int tnr = (tnr4 >> (5 * t)) & 31;      // valid rows (the last tile of a neighborhood is partial)
        float* xw = reinterpret_cast<float*>(lds + (trow << 8) + ((r & 3) << 2));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float sm = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = fmaxf(acc[t][u][e], 0.f);
            const int rl = 4 * q + e;                  // row in the tile = its low 4 bits (tiles are 16-aligned)
            if (!last) xw[(rl << 6) + ((((4 * u + (r >> 2)) ^ rl) & 15) << 2)] = v;
            sm += (rl < tnr) ? v : 0.f;
          }
          // column sums over the four row groups (lane quarters): quarter pairs, then halves; fixed order
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          u32x2 w_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
          sm = __uint_as_float(w_[0]) + __uint_as_float(w_[1]);
          w_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
          sm = __uint_as_float(w_[0]) + __uint_as_float(w_[1]);
          if (q == 0) Pl[RES_TK(t) * 64 + 16 * u + r] = sm;
        }
      }
      if (wave == 7) {
        float* xw = reinterpret_cast<float*>(lds + ((r & 3) << 2));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = fmaxf(acc[3][u][e], 0.f);
            const int rl = 4 * q + e;
            if (!last) xw[(rl << 6) + ((((4 * u + (r >> 2)) ^ rl) & 15) << 2)] = v;
            if (rl < nnb) a.canon[(int64_t)info->nb[rl] * a.ldc + (L + 2) * 64 + 16 * u + r] = v;
          }
      }
      RES_T(6)
      __syncthreads();
      RES_T(7)
      RES_POOL_OUT(L + 2)
      RES_T(8)
    }
    __syncthreads();      // the partials and `info` are rewritten by the next pack
  }
#ifdef RES_PROF
  __syncthreads();
  if (tid < NWAVE * 10) atomicAdd(&res_prof[(tid < 10 ? 0 : 16) + tid % 10], (unsigned long long)profl[tid]);
#endif
}

}  // namespace res

// ---- host side -----------------------------------------------------------------------------------------
static int device_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cus;
}

#ifdef RES_PROF
extern "C" int desco_debug_resident_prof(unsigned long long* out, int reset) {
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(res::res_prof), sizeof(unsigned long long) * 32);
  if (reset) {
    unsigned long long z[32] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(res::res_prof), z, sizeof(z));
  }
  return 0;
}
#endif

extern "C" int desco_shmp_resident_limits(int* max_count_rows, int* max_edges, int* max_neigh) {
  if (max_count_rows) *max_count_rows = res::MAXCT * 16;
  if (max_edges) *max_edges = res::ECAP;
  if (max_neigh) *max_neigh = res::MAXNB;
  return 0;
}

extern "C" int desco_resident_plan(const int32_t* count_ptr, const int32_t* vrowptr, int64_t num_neigh,
                                   int64_t num_count, int min_count_rows, uint8_t* eligible,
                                   int32_t* pack_list, int64_t* num_packs) {
  if (!count_ptr || !vrowptr || !eligible || !pack_list || !num_packs || num_neigh < 0 || min_count_rows < 1)
    return fail(DESCO_EINVAL, "desco_resident_plan: bad argument");
  // Bin packing by tiles (first fit, largest first): every pack is filled towards the kernel's 30 count
  // tiles, so that its eight waves have the same number of tiles and the per-pack fixed costs are paid as
  // rarely as possible.  A neighborhood's result does not depend on its pack (see the kernel header).
  std::vector<int32_t> tiles((size_t)num_neigh), edges((size_t)num_neigh);
  std::vector<std::vector<int32_t>> bucket(res::MAXCT + 1);
  for (int64_t b = 0; b < num_neigh; ++b) {
    const int64_t c0 = count_ptr[b], c1 = count_ptr[b + 1];
    const int64_t n = c1 - c0;
    const int64_t e = ((int64_t)vrowptr[4 * c1] - vrowptr[4 * c0]) +
                      ((int64_t)vrowptr[4 * (num_count + b) + 4] - vrowptr[4 * (num_count + b)]);
    const int64_t t = (n + 15) / 16;
    // eligibility is a property of the neighborhood alone: the same choice in every shard / block
    const bool ok = n >= min_count_rows && t <= res::MAXCT && e <= res::ECAP;
    eligible[b] = ok ? 1 : 0;
    if (ok) {
      tiles[(size_t)b] = (int32_t)t;
      edges[(size_t)b] = (int32_t)e;
      bucket[(size_t)t].push_back((int32_t)b);
    }
  }
  std::vector<size_t> head(res::MAXCT + 1, 0);      // buckets are consumed front to back (ascending index)
  int64_t np = 0;
  for (int top = res::MAXCT; top >= 1;) {
    if (head[top] >= bucket[top].size()) {
      --top;
      continue;
    }
    int32_t* pl = pack_list + np * res::MAXNB;
    for (int j = 0; j < res::MAXNB; ++j) pl[j] = -1;
    int nnb = 0, ct = 0, ce = 0;
    for (int sz = top; sz >= 1 && nnb < res::MAXNB; --sz) {
      while (nnb < res::MAXNB && head[sz] < bucket[sz].size() && ct + sz <= res::MAXCT) {
        const int32_t b = bucket[sz][head[sz]];
        if (ce + edges[(size_t)b] > res::ECAP) break;       // (try smaller neighborhoods)
        pl[nnb++] = b;
        ct += sz;
        ce += edges[(size_t)b];
        ++head[sz];
      }
    }
    ++np;
  }
  *num_packs = np;
  return 0;
}

extern "C" int desco_shmp_resident_bf16x6_f32(const int32_t* count_ptr, const int32_t* vrowptr,
                                              const int32_t* vcol, int64_t num_count,
                                              const int32_t* pack_list, int num_packs, const float* l0coef, const int16_t* wfrag,
                                              const float* bias, int num_layers, float* pooled,
                                              int64_t ldp, float* canon, int64_t ldc, int32_t* counter,
                                              desco_stream_t stream) {
  if (!count_ptr || !vrowptr || !vcol || !pack_list || !l0coef || !pooled || !canon || !counter ||
      num_packs < 0 || num_layers < 0 || (num_layers > 0 && (!wfrag || !bias)) ||
      ldp < 64 * (num_layers + 2) || ldc < 64 * (num_layers + 2))
    return fail(DESCO_EINVAL, "desco_shmp_resident_bf16x6_f32: bad argument");
  if (num_packs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  static DeviceOnce attr_once;
  if (!attr_once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(res::shmp_resident_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.mark();
  }
  hipError_t e = hipMemsetAsync(counter, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return fail((int)e, "desco_shmp_resident_bf16x6_f32: hipMemsetAsync failed");
  res::Args a;
  a.count_ptr = count_ptr;
  a.vrowptr = vrowptr;
  a.vcol = vcol;
  a.num_count = num_count;
  a.pack_list = pack_list;
  a.num_packs = num_packs;
  a.l0coef = l0coef;
  a.wfrag = reinterpret_cast<const char*>(wfrag);
  a.bias = bias;
  a.num_layers = num_layers;
  a.pooled = pooled;
  a.ldp = ldp;
  a.canon = canon;
  a.ldc = ldc;
  a.counter = counter;
  const int cus = device_cus();
  const unsigned grid = (unsigned)(num_packs < cus ? num_packs : cus);
  hipLaunchKernelGGL(res::shmp_resident_kernel, dim3(grid), dim3(res::NWAVE * 64), res::LDS_BYTES, st, a);
  return launch_status("desco_shmp_resident_bf16x6_f32");
}

}  // namespace desco
