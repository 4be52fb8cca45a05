// Fused SHMP layer, 16-row wave tiles, gathers issued ONE TILE AHEAD (the product path's kernel).
//
// Same layer and arithmetic as shmp_layer.hip / shmp_layer16.hip (gnn_model.py:47-70 of the reference:
// per relation SAGEConv(aggr=add) neighbor sums, root/self Linear, HeteroConv sum, relu; bf16 6-product
// MFMAs at fp32 accuracy), and the same wave-autonomous tiles of 16 rows.  What changes is when the
// gathers are issued.
//
// Why.  The kernel is bound by the latency of its gathers, not by HBM bandwidth or by the matrix pipe
// (profiles/r2_h: matrix pipe 14 % busy, waves waiting 50-70 % of their cycles, 3.1 TB/s of the
// ~5.5 TB/s a pure row gather reaches).  By Little's law the rate is (useful bytes in flight) / (loaded
// latency, ~3.4 us): the block-pipelined kernels keep ONE gather step (2 sources x 16 rows = 8 KB per
// wave) in flight and only for the time of one block's MFMAs.  Registers that hold loads in flight are
// what buys bandwidth, and more waves do not add any: 16 waves x 128 registers leave 32 per wave for
// that, 8 waves x 256 leave 128.
//
// Here a wave keeps the first gather step of EVERY block of its next tile in flight (relation slots,
// self rows, table slots: up to 28 KB per wave, 8 waves per CU) while it works on the current tile:
//   tile i, block b:  consume G_b (loaded during tile i-1; + extra steps / cooperative path for heavy
//                     rows, using the same registers)  ->  issue block b of tile i+1 into G_b  ->
//                     split / LDS / MFMA of block b.
// Every first step has a whole tile of work (~2 us with two waves per SIMD) to arrive.  The CSR slices
// (row pointers, source ids) run one tile further ahead, global -> LDS directly (global_load_lds_dword),
// in three buffers: tile i (cursors of the extra steps), tile i+1 (addresses of the gathers being
// issued), tile i+2 (in flight).  Vector memory returns in order, so one wait on the oldest gather
// register of the tile (block 0, issued right after the CSR loads of the previous iteration) also
// proves that those have landed: no counted s_waitcnt, no vmcnt(0) anywhere in the loop.  The first and
// last rows of tile i+2's id slice come from two scalar loads a tile earlier.
//
// Dead relation slots (no source among the tile's 16 rows: triangle edges in molecule graphs) are
// neither gathered nor multiplied, except block 0, whose loads carry the ordering above.
#include "common_device.hpp"
#include "shmp_args.hpp"

namespace desco {

constexpr int WR = 16;        // rows per wave
constexpr int NW = 8;         // waves per block (2 per SIMD, 256 registers each)
constexpr int AH = 36;        // half-K fp32 table image row stride (floats): conflict-free ds_read_b32 / ds_write_b128
constexpr int APS = 32;       // half-K bf16 plane row stride (shorts, 64 B), chunks XOR-swizzled
constexpr int MAXS = 4;       // relation slots stored per row
constexpr int RPN = 128;      // row-pointer buffer (16 * MAXS + 1 used; both LDS-direct chunks are whole)
constexpr int EXTRA_STEPS = 9; // batched 2-source steps after the prefetched one (<= 20 sources per row)
constexpr int WCAP = 384;     // source ids staged per wave and buffer (longer slices fall back to global)
constexpr int A_FLOATS = 3 * WR * APS / 2;               // A region per wave: max(16*36, 3*16*32/2) floats
constexpr int WAVE_LDS = A_FLOATS + 3 * RPN + 3 * WCAP;  // floats per wave
static_assert(A_FLOATS >= WR * AH, "the fp32 image must fit in the plane region");
static_assert(WCAP % 64 == 0, "ids are staged in chunks of 64");

// absent sources of a batched gather step read this row instead of being predicated away
__device__ __attribute__((aligned(16))) float shmp16t_zero_row[64] = {};

using bf16x8 = __attribute__((ext_vector_type(8))) short;

__device__ __forceinline__ void f4add(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

// one gather step of one block in flight: two sources (u, w) of rows g8 and 8 + g8, low / high half row
struct Gath {
  float4 u00, u01, u10, u11, w00, w01, w10, w11;
};

// 4 bytes per lane from global straight into LDS: lane i's dword lands at dst_[i] (dst_ wave-uniform).
// Completion is counted by vmcnt like any vector load, but the compiler does not order later LDS reads
// behind it (see the ordering note in the header).
#define DESCO_DMA4(src_, dst_)                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_),      \
                                   (__attribute__((address_space(3))) void*)(dst_), 4, 0, 0)

// ---- gather machinery (macros: every temporary is a named register) ---------------------------------
// X_ = C (current tile) or N (next tile) selects rp##X_, ec##X_, ebase##X_, grow##X_, nr##X_.
// first step of relation slot slot_ for row it_: two sources, staged ids only, unconditional loads
#define DESCO_ISSUE_SLOT(G_, it_, slot_, X_)                                                   \
  {                                                                                            \
    const int v_ = ((it_) * 8 + g8) * S + (slot_);                                             \
    const int c_ = rp##X_[v_] - ebase##X_, n_ = rp##X_[v_ + 1] - ebase##X_;                    \
    const int m_ = nr##X_ > 0 ? (n_ < WCAP ? n_ : WCAP) : c_;                                  \
    const bool k0_ = c_ < m_, k1_ = c_ + 1 < m_;                                               \
    const int i0_ = ec##X_[k0_ ? c_ : 0], i1_ = ec##X_[k1_ ? c_ + 1 : 0];                      \
    const float* p0_ = k0_ ? xb + (int64_t)i0_ * LDX : zrow;                                   \
    const float* p1_ = k1_ ? xb + (int64_t)i1_ * LDX : zrow;                                   \
    G_.u##it_##0 = *reinterpret_cast<const float4*>(p0_);                                      \
    G_.u##it_##1 = *reinterpret_cast<const float4*>(p0_ + 32);                                 \
    G_.w##it_##0 = *reinterpret_cast<const float4*>(p1_);                                      \
    G_.w##it_##1 = *reinterpret_cast<const float4*>(p1_ + 32);                                 \
  }
// the row itself: rows beyond nr re-read the wave's last valid row (never stored)
#define DESCO_ISSUE_SELF(G_, it_, X_)                                                          \
  {                                                                                            \
    const int r_ = (it_) * 8 + g8;                                                             \
    const float* p_ = nr##X_ > 0 ? xb + (grow##X_ + (r_ < nr##X_ ? r_ : nr##X_ - 1)) * LDX : zrow; \
    G_.u##it_##0 = *reinterpret_cast<const float4*>(p_);                                       \
    G_.u##it_##1 = *reinterpret_cast<const float4*>(p_ + 32);                                  \
  }
// table pseudo block: the first source of table slot 0 (-> u) and of table slot 1 (-> w) of row it_
#define DESCO_TAB_CUR(it_, X_)                                                              \
  const int v_ = ((it_) * 8 + g8) * S + g.sm;                                               \
  const int ca_ = rp##X_[v_] - ebase##X_, na_ = rp##X_[v_ + 1] - ebase##X_;                 \
  const int nb_ = ST > 1 ? rp##X_[v_ + 2] - ebase##X_ : na_;                                \
  const bool k0_ = nr##X_ > 0 && ca_ < (na_ < WCAP ? na_ : WCAP);                           \
  const bool k1_ = ST > 1 && nr##X_ > 0 && na_ < (nb_ < WCAP ? nb_ : WCAP);
#define DESCO_ISSUE_TAB(G_, it_, X_)                                                        \
  {                                                                                         \
    DESCO_TAB_CUR(it_, X_)                                                                  \
    const int i0_ = ec##X_[k0_ ? ca_ : 0], i1_ = ec##X_[k1_ ? na_ : 0];                     \
    const float* p0_ = k0_ ? yb + (int64_t)i0_ * LDY : zrow;                                \
    G_.u##it_##0 = *reinterpret_cast<const float4*>(p0_);                                   \
    G_.u##it_##1 = *reinterpret_cast<const float4*>(p0_ + 32);                              \
    if (ST > 1) {                                                                           \
      const float* p1_ = k1_ ? yb + 64 + (int64_t)i1_ * LDY : zrow;                         \
      G_.w##it_##0 = *reinterpret_cast<const float4*>(p1_);                                 \
      G_.w##it_##1 = *reinterpret_cast<const float4*>(p1_ + 32);                            \
    }                                                                                       \
  }
// first step of block b_ of tile X_ (nothing waits on the loads here)
#define DESCO_ISSUE_BLOCK(b_, G_, X_)                                                      \
  {                                                                                        \
    if ((b_) < KB - 1) {                                                                   \
      DESCO_ISSUE_SLOT(G_, 0, b_, X_) DESCO_ISSUE_SLOT(G_, 1, b_, X_)                      \
    } else if ((b_) == KB - 1) {                                                           \
      DESCO_ISSUE_SELF(G_, 0, X_) DESCO_ISSUE_SELF(G_, 1, X_)                              \
    } else {                                                                               \
      DESCO_ISSUE_TAB(G_, 0, X_) DESCO_ISSUE_TAB(G_, 1, X_)                                \
    }                                                                                      \
  }
// ---- the current tile (rp, ec, ebase = its buffers) ---------------------------------------------------
// cursor [c, n) of relation slot slot_ of row it_ BEHIND the first step (issued a tile ago)
#define DESCO_CUR(it_, slot_)                                                           \
  {                                                                                     \
    const int v_ = ((it_) * 8 + g8) * S + (slot_);                                      \
    c##it_ = rp[v_] - ebase;                                                            \
    n##it_ = rp[v_ + 1] - ebase;                                                        \
    const int m_ = n##it_ < WCAP ? n##it_ : WCAP;                                       \
    c##it_ += (c##it_ < m_ ? 1 : 0) + (c##it_ + 1 < m_ ? 1 : 0);                        \
  }
// two more sources of row it_ (staged ids only: e < WCAP), unconditional loads
#define DESCO_ISSUE2(G_, it_, base_, ld_)                                             \
  {                                                                                   \
    const int m_ = n##it_ < WCAP ? n##it_ : WCAP;                                     \
    const bool k0_ = c##it_ < m_, k1_ = c##it_ + 1 < m_;                              \
    const int i0_ = ec[k0_ ? c##it_ : 0], i1_ = ec[k1_ ? c##it_ + 1 : 0];             \
    const float* p0_ = k0_ ? (base_) + (int64_t)i0_ * (ld_) : zrow;                   \
    const float* p1_ = k1_ ? (base_) + (int64_t)i1_ * (ld_) : zrow;                   \
    G_.u##it_##0 = *reinterpret_cast<const float4*>(p0_);                             \
    G_.u##it_##1 = *reinterpret_cast<const float4*>(p0_ + 32);                        \
    G_.w##it_##0 = *reinterpret_cast<const float4*>(p1_);                             \
    G_.w##it_##1 = *reinterpret_cast<const float4*>(p1_ + 32);                        \
    c##it_ += (k0_ ? 1 : 0) + (k1_ ? 1 : 0);                                          \
  }
#define DESCO_CONSUME2(G_, it_)                                                       \
  {                                                                                   \
    f4add(lo##it_, G_.u##it_##0);                                                     \
    f4add(hi##it_, G_.u##it_##1);                                                     \
    f4add(lo##it_, G_.w##it_##0);                                                     \
    f4add(hi##it_, G_.w##it_##1);                                                     \
  }
#define DESCO_ZERO_SUMS()                                   \
  {                                                         \
    lo0 = make_float4(0.f, 0.f, 0.f, 0.f);                  \
    lo1 = lo0;                                              \
    hi0 = lo0; hi1 = lo0;                                   \
  }
#define DESCO_ANY_STAGED()                                                                    \
  __any((c0 < (n0 < WCAP ? n0 : WCAP)) | (c1 < (n1 < WCAP ? n1 : WCAP)))
// consume the table step; leave the cursor of table slot 0 in (c, n) and of slot 1 in (d, m)
#define DESCO_CONSUME_TAB(G_, it_)                                                          \
  {                                                                                         \
    DESCO_TAB_CUR(it_, C)                                                                   \
    f4add(lo##it_, G_.u##it_##0);                                                           \
    f4add(hi##it_, G_.u##it_##1);                                                           \
    if (ST > 1) {                                                                           \
      f4add(lo##it_, G_.w##it_##0);                                                         \
      f4add(hi##it_, G_.w##it_##1);                                                         \
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
// finish a relation-slot block whose first step was issued a tile ago: consume it, up to EXTRA_STEPS
// more batched steps (two sources per row each, in the block's own registers), then the cooperative
// path for rows that are heavier still (one row at a time, the whole wave on it)
#define DESCO_FINISH(G_, slot_)                                                            \
  {                                                                                        \
    DESCO_CUR(0, slot_) DESCO_CUR(1, slot_)                                                \
    DESCO_CONSUME2(G_, 0) DESCO_CONSUME2(G_, 1)                                            \
    for (int st_ = 0; st_ < EXTRA_STEPS && DESCO_ANY_STAGED(); ++st_) {                    \
      DESCO_ISSUE2(G_, 0, xb, LDX) DESCO_ISSUE2(G_, 1, xb, LDX)                            \
      DESCO_CONSUME2(G_, 0) DESCO_CONSUME2(G_, 1)                                          \
    }                                                                                      \
    if (__any((c0 < n0) | (c1 < n1))) {                                                    \
      DESCO_COOP(0, xb, LDX) DESCO_COOP(1, xb, LDX)                                        \
    }                                                                                      \
  }
// bit s of the live mask of tile X_: relation slot s has at least one source among the wave's 16 rows
#define DESCO_SLOT_LIVE(s_, X_)                                                                        \
  (__any((rp##X_[(0 * 8 + g8) * S + (s_) + 1] > rp##X_[(0 * 8 + g8) * S + (s_)]) |                     \
         (rp##X_[(1 * 8 + g8) * S + (s_) + 1] > rp##X_[(1 * 8 + g8) * S + (s_)]))                      \
       ? 1 << (s_) : 0)
#define DESCO_TILE_LIVE(live_, X_)                                        \
  {                                                                       \
    live_ = 0;                                                            \
    if (nr##X_ > 0) {                                                     \
      if (KB - 1 > 0) live_ |= DESCO_SLOT_LIVE(0, X_);                    \
      if (KB - 1 > 1) live_ |= DESCO_SLOT_LIVE(1, X_);                    \
    }                                                                     \
  }
// write one fp32 half image (every lane writes: row = it*8 + g8, 4 floats at 4*l8)
#define DESCO_PUT_F32(av_, it_)                         \
  {                                                     \
    *reinterpret_cast<float4*>(Aw + ((it_) * 8 + g8) * AH + 4 * l8) = av_;   \
  }
// write one half image as three bf16 planes (row = it*8 + g8, 4 bf16 at 4*l8 of every plane).  Plane
// rows are 64 B with no padding: the 16-byte chunk k/8 of row r sits at chunk (k/8) ^ (-(r/4) & 3), which
// makes both this write (ds_write_b64) and the fragment read conflict-free (ds_read_b128 is served in
// the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: tools/micro/lds_banks.py)
#define DESCO_PUT_X6(av_, it_)                                                  \
  {                                                                             \
    uint32_t h0_, m0_, l0_, h1_, m1_, l1_;                                      \
    split2_bf16x3(av_.x, av_.y, h0_, m0_, l0_);                                   \
    split2_bf16x3(av_.z, av_.w, h1_, m1_, l1_);                                   \
    short* d_ = Ap + ((it_) * 8 + g8) * APS +                                   \
                ((((l8 >> 1) ^ (0 - ((it_) * 2 + (g8 >> 2)))) & 3) << 3) + 4 * (l8 & 1); \
    *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                       \
    *reinterpret_cast<uint2*>(d_ + WR * APS) = make_uint2(m0_, m1_);            \
    *reinterpret_cast<uint2*>(d_ + 2 * WR * APS) = make_uint2(l0_, l1_);        \
  }
// 24 bf16 MFMAs (6-product split) of v_mfma_f32_16x16x32_bf16 on the staged half (32 k) of block b_: lane
// (r = lane&15, q = lane>>4) holds A[row r][k = 8 q + 0..7] and B[k = 8 q + 0..7][col 16 t + r] of every
// plane, t = 0..3 (the four 16-column tiles of the 64 outputs)
#define DESCO_M16(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0);
#define DESCO_MFMA_HALF_X6(b_, h_)                                                                \
  {                                                                                               \
    const short* ap_ = Ap + (lane & 15) * APS + ((((lane >> 4) ^ (0 - (lane >> 2))) & 3) << 3);   \
    const short* bp_ = Wp + (lane & 15) * WST + (b_) * 64 + (h_) * 32 + 8 * (lane >> 4);          \
    const bf16x8 ah_ = *reinterpret_cast<const bf16x8*>(ap_);                                     \
    const bf16x8 am_ = *reinterpret_cast<const bf16x8*>(ap_ + WR * APS);                          \
    const bf16x8 al_ = *reinterpret_cast<const bf16x8*>(ap_ + 2 * WR * APS);                      \
    const bf16x8 b0h_ = *reinterpret_cast<const bf16x8*>(bp_);                                    \
    const bf16x8 b0m_ = *reinterpret_cast<const bf16x8*>(bp_ + WPL);                              \
    const bf16x8 b0l_ = *reinterpret_cast<const bf16x8*>(bp_ + 2 * WPL);                          \
    const bf16x8 b1h_ = *reinterpret_cast<const bf16x8*>(bp_ + 16 * WST);                         \
    const bf16x8 b1m_ = *reinterpret_cast<const bf16x8*>(bp_ + 16 * WST + WPL);                   \
    const bf16x8 b1l_ = *reinterpret_cast<const bf16x8*>(bp_ + 16 * WST + 2 * WPL);               \
    DESCO_M16(al_, b0h_, q0) DESCO_M16(al_, b1h_, q1)                                             \
    DESCO_M16(ah_, b0l_, q0) DESCO_M16(ah_, b1l_, q1)                                             \
    DESCO_M16(am_, b0m_, q0) DESCO_M16(am_, b1m_, q1)                                             \
    const bf16x8 b2h_ = *reinterpret_cast<const bf16x8*>(bp_ + 32 * WST);                         \
    const bf16x8 b2m_ = *reinterpret_cast<const bf16x8*>(bp_ + 32 * WST + WPL);                   \
    const bf16x8 b2l_ = *reinterpret_cast<const bf16x8*>(bp_ + 32 * WST + 2 * WPL);               \
    const bf16x8 b3h_ = *reinterpret_cast<const bf16x8*>(bp_ + 48 * WST);                         \
    const bf16x8 b3m_ = *reinterpret_cast<const bf16x8*>(bp_ + 48 * WST + WPL);                   \
    const bf16x8 b3l_ = *reinterpret_cast<const bf16x8*>(bp_ + 48 * WST + 2 * WPL);               \
    DESCO_M16(am_, b0h_, q0) DESCO_M16(am_, b1h_, q1)                                             \
    DESCO_M16(ah_, b0m_, q0) DESCO_M16(ah_, b1m_, q1)                                             \
    DESCO_M16(ah_, b0h_, q0) DESCO_M16(ah_, b1h_, q1)                                             \
    DESCO_M16(al_, b2h_, q2) DESCO_M16(al_, b3h_, q3)                                             \
    DESCO_M16(ah_, b2l_, q2) DESCO_M16(ah_, b3l_, q3)                                             \
    DESCO_M16(am_, b2m_, q2) DESCO_M16(am_, b3m_, q3)                                             \
    DESCO_M16(am_, b2h_, q2) DESCO_M16(am_, b3h_, q3)                                             \
    DESCO_M16(ah_, b2m_, q2) DESCO_M16(ah_, b3m_, q3)                                             \
    DESCO_M16(ah_, b2h_, q2) DESCO_M16(ah_, b3h_, q3)                                             \
  }
// add the staged (fp32) table half rows (32 columns) in the C/D layout of two 16-column tiles:
// lane (c = lane&15, g = lane>>4) holds rows 4 g + e, column 16 t + c
#define DESCO_TAB_HALF(qa_, qb_)                                                            \
  {                                                                                        \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                     \
      const float* t_ = Aw + (4 * (lane >> 4) + e_) * AH + (lane & 15);                    \
      qa_[e_] += t_[0];                                                                    \
      qb_[e_] += t_[16];                                                                   \
    }                                                                                      \
  }

// one K block of the current tile + the same block of the next tile: b_ < KB-1 = relation slot b_
// (gathered x rows), b_ == KB-1 = the row itself, b_ == KB (ST > 0) = table pseudo block (gathered ytab
// rows, added in the C/D layout).  Relation slots other than 0 that are empty for the tile are skipped.
#define DESCO_BLOCK(b_, G_)                                                                          \
  if ((b_) < NB) {                                                                                   \
    const bool cons_ = (b_) == 0 || (b_) >= KB - 1 || ((liveC >> (b_)) & 1);                         \
    const bool mul_ = (b_) >= KB - 1 || ((liveC >> (b_)) & 1);                                       \
    const bool issue_ = (b_) == 0 || (b_) >= KB - 1 || ((liveN >> (b_)) & 1);                        \
    if (cons_) {                                                                                     \
      DESCO_ZERO_SUMS()                                                                              \
      if ((b_) < KB - 1) {                                                                           \
        DESCO_FINISH(G_, b_)                                                                         \
      } else if ((b_) == KB - 1) {                                                                   \
        lo0 = G_.u00; hi0 = G_.u01; lo1 = G_.u10; hi1 = G_.u11;                                      \
      } else {                                                                                       \
        /* canonical->count relations have at most one source per row: one step covers both table */ \
        /* slots; anything beyond that (general inputs) takes the cooperative path */                \
        int d0, d1, m0, m1;                                                                          \
        DESCO_CONSUME_TAB(G_, 0) DESCO_CONSUME_TAB(G_, 1)                                            \
        if (__any((c0 < n0) | (c1 < n1))) {                                                          \
          DESCO_COOP(0, yb, LDY) DESCO_COOP(1, yb, LDY)                                              \
        }                                                                                            \
        if (ST > 1 && __any((d0 < m0) | (d1 < m1))) {                                                \
          c0 = d0; c1 = d1;                                                                          \
          n0 = m0; n1 = m1;                                                                          \
          DESCO_COOP(0, yb + 64, LDY) DESCO_COOP(1, yb + 64, LDY)                                    \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
    if (issue_) DESCO_ISSUE_BLOCK(b_, G_, N)                                                         \
    if (mul_) {                                                                                      \
      if ((b_) < KB) {                                                                               \
        DESCO_PUT_X6(lo0, 0) DESCO_PUT_X6(lo1, 1)                                                    \
        DESCO_MFMA_HALF_X6(b_, 0)                                                                    \
        DESCO_PUT_X6(hi0, 0) DESCO_PUT_X6(hi1, 1)                                                    \
        DESCO_MFMA_HALF_X6(b_, 1)                                                                    \
      } else {                                                                                       \
        DESCO_PUT_F32(lo0, 0) DESCO_PUT_F32(lo1, 1)                                                  \
        DESCO_TAB_HALF(q0, q1)                                                                       \
        DESCO_PUT_F32(hi0, 0) DESCO_PUT_F32(hi1, 1)                                                  \
        DESCO_TAB_HALF(q2, q3)                                                                       \
      }                                                                                              \
    }                                                                                                \
  }
// rows of tile t_ that belong to this wave (nr_ = 0: none)
#define DESCO_TILE_ROWS(t_, w0_, nr_)                                                     \
  {                                                                                       \
    w0_ = (t_) * (NW * WR) + wave * WR;                                                   \
    nr_ = ((t_) < ntiles && w0_ < g.num_rows)                                             \
              ? (int)((g.num_rows - w0_) < WR ? (g.num_rows - w0_) : WR) : 0;             \
  }

// KB = sm + 1 resident weight blocks (1..3), ST table slots (0..2),
// LD64: x rows are 64 floats and ytab rows 64*ST floats apart (the product path's layouts): source-row
// addresses then need a shift instead of a 64-bit multiply per gathered row
// POOL: fused pooling epilogue (instantiated for the count-row launches only)
template <int KB, int ST, bool LD64, bool POOL>
__global__ __launch_bounds__(NW * 64) void shmp_layer16t_kernel(ShmpArgs g, const int32_t* __restrict__ rowptr_s,
                                                               const uint32_t* __restrict__ pool_bits_s,
                                                               const int32_t* __restrict__ pool_slot_s) {
  // rowptr_s / pool_*_s = g.vrowptr / g.pool_bits / g.pool_slot once more, as read-only restrict
  // parameters: their wave-uniform loads then go through the scalar cache (s_load, counted by lgkmcnt)
  // instead of queueing behind the gathers in the in-order vector memory pipe
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WST = KB * 64 + 16;                        // weight plane row stride (shorts): 32 B of padding, conflict-free B fragments
  constexpr int WPL = 64 * WST;                            // shorts per weight plane
  constexpr int W_FLOATS = 3 * WPL / 2;
  short* Wp = reinterpret_cast<short*>(lds);               // [3][64 n][WST]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile bookkeeping stays in SGPRs
  float* Aw = lds + W_FLOATS + wave * WAVE_LDS;            // fp32 half image [16][36] (table block)
  short* Ap = reinterpret_cast<short*>(Aw);                // bf16 planes [3][16][32] of a half image
  int* rpb = reinterpret_cast<int*>(Aw + A_FLOATS);        // 3 x [RPN] row pointers (absolute)
  int* ecb = rpb + 3 * RPN;                                // 3 x [WCAP] source ids

  // ---- resident weights -------------------------------------------------------------------
  {
    // global planes [3][64][KB*64] -> LDS [3][64][WST], 16 bytes at a time
    constexpr int CH = KB * 8;                             // uint4 chunks per row
    for (int i = tid; i < 3 * 64 * CH; i += NW * 64) {
      const int row = i / CH, ch = i - row * CH;           // row = plane*64 + n
      *reinterpret_cast<uint4*>(Wp + row * WST + 8 * ch) =
          *reinterpret_cast<const uint4*>(g.wplanes + (int64_t)row * (KB * 64) + 8 * ch);
    }
  }
  __syncthreads();

  const int g8 = lane >> 3, l8 = lane & 7;                 // 8 groups of 8 lanes: one half row each
  const int S = g.S;
  const int nslot = WR * S + 1;                            // <= 65
  const int64_t ntiles = (g.num_rows + NW * WR - 1) / (NW * WR);
  constexpr int NB = KB + (ST > 0 ? 1 : 0);                // K blocks incl. the table pseudo block
  const int64_t LDX = LD64 ? 64 : g.ldx, LDY = LD64 ? 64 * (ST > 0 ? ST : 1) : g.ldy;
  const float* xb = g.x + 4 * l8;
  const float* yb = ST > 0 ? g.ytab + 4 * l8 - g.ytab_row0 * LDY : nullptr;
  const float* zrow = shmp16t_zero_row + 4 * l8;
  (void)yb;

  // ---- tile order (XCD-aware, see shmp_layer.hip): chunks of (grid / 8) consecutive tiles go round
  //      robin over the XCDs -----------------------------------------------------------------------
  int64_t tile;
  const int tstride = gridDim.x;
  if ((gridDim.x & 7) == 0)
    tile = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  else
    tile = blockIdx.x;

  int64_t w0C, w0N, w0F;                                   // first row (relative) of the current / next / far tile
  int nrC, nrN, nrF;
  DESCO_TILE_ROWS(tile, w0C, nrC)
  if (nrC == 0) return;                                    // no barrier below: idle waves may leave
  DESCO_TILE_ROWS(tile + tstride, w0N, nrN)
  DESCO_TILE_ROWS(tile + 2 * tstride, w0F, nrF)
  int64_t growC = g.row0 + w0C, growN = g.row0 + w0N;
  int* rpC = rpb;
  int* rpN = rpb + RPN;
  int* rpF = rpb + 2 * RPN;
  int* ecC = ecb;
  int* ecN = ecb + WCAP;
  int* ecF = ecb + 2 * WCAP;

  // ---- prologue: CSR slices of the first two tiles through registers, scalars of the third --------
  int ebaseC = 0, ebaseN = 0;
  if (S > 0) {
    {
      const int nptr = nrC * S + 1;
      for (int i = lane; i < nslot; i += 64) rpC[i] = g.vrowptr[growC * S + (i < nptr ? i : nptr - 1)];
      ebaseC = __builtin_amdgcn_readfirstlane(rpC[0]);
      const int ecnt = __builtin_amdgcn_readfirstlane(rpC[WR * S]) - ebaseC;
      for (int i = lane; i < ecnt && i < WCAP; i += 64) ecC[i] = g.vcol[ebaseC + i];
    }
    if (nrN > 0) {
      const int nptr = nrN * S + 1;
      for (int i = lane; i < nslot; i += 64) rpN[i] = g.vrowptr[growN * S + (i < nptr ? i : nptr - 1)];
      const int eb = __builtin_amdgcn_readfirstlane(rpN[0]);
      const int ecnt = __builtin_amdgcn_readfirstlane(rpN[WR * S]) - eb;
      for (int i = lane; i < ecnt && i < WCAP; i += 64) ecN[i] = g.vcol[eb + i];
    }
  }
  int ebF = 0, eeF = 0;                                    // id range of the far tile (scalar loads)
  if (S > 0 && nrF > 0) {
    const int32_t* q_ = rowptr_s + (g.row0 + w0F) * S;
    ebF = q_[0];
    eeF = q_[nrF * S];
  }
  // bias of the lane's four output columns (C/D layout), once
  float bq0 = 0.f, bq1 = 0.f, bq2 = 0.f, bq3 = 0.f;
  if (g.bias) {
    bq0 = g.bias[lane & 15];
    bq1 = g.bias[16 + (lane & 15)];
    bq2 = g.bias[32 + (lane & 15)];
    bq3 = g.bias[48 + (lane & 15)];
  }

  float4 lo0, lo1, hi0, hi1;                               // gathered sums of the current block
  int c0 = 0, c1 = 0, n0 = 0, n1 = 0;                      // cursors [c, n) rel. to ebase (extra steps)
  Gath G0, G1, G2, G3;                                     // first gather step of every block, one tile ahead
  int liveC = 0, liveN = 0;
  {
    int* rp = rpC;
    (void)rp;
    DESCO_TILE_LIVE(liveC, C)
    // (the N-form issue macros read rpN/ecN/...: alias the first tile as "next" for this one use)
  }
  {
    // issue every block of the first tile
    int* rpS = rpN; int* ecS = ecN; const int ebaseS = ebaseN; const int64_t growS = growN; const int nrS = nrN;
    rpN = rpC; ecN = ecC; ebaseN = ebaseC; growN = growC; nrN = nrC;
    if (0 < NB) DESCO_ISSUE_BLOCK(0, G0, N)
    if (1 < NB && (1 >= KB - 1 || ((liveC >> 1) & 1))) DESCO_ISSUE_BLOCK(1, G1, N)
    if (2 < NB) DESCO_ISSUE_BLOCK(2, G2, N)
    if (3 < NB) DESCO_ISSUE_BLOCK(3, G3, N)
    rpN = rpS; ecN = ecS; ebaseN = ebaseS; growN = growS; nrN = nrS;
  }

  for (;;) {
    int* rp = rpC;                                         // (names used by the current-tile macros)
    int* ec = ecC;
    const int ebase = ebaseC;
    (void)rp; (void)ec; (void)ebase;
    // ---- the oldest gather of this tile has landed => so has everything issued before it: the CSR
    //      slices of the NEXT tile (LDS-direct loads of the previous iteration).  The asm ties the wait
    //      to that register and keeps the LDS reads below it.
    asm volatile("" : "+v"(G0.u00.x) : : "memory");
    if (S > 0 && nrN > 0) ebaseN = __builtin_amdgcn_readfirstlane(rpN[0]);
    DESCO_TILE_LIVE(liveN, N)
    // ---- CSR slices of the far tile (i + 2): global -> LDS, no staging registers ------------------
    if (S > 0 && nrF > 0) {
      const int nptr = nrF * S + 1;
      const int32_t* src = g.vrowptr + (g.row0 + w0F) * S;
      DESCO_DMA4(src + (lane < nptr ? lane : nptr - 1), rpF);
      if (nslot > 64) DESCO_DMA4(src + (lane + 64 < nptr ? lane + 64 : nptr - 1), rpF + 64);
      const int ne = (eeF - ebF) < WCAP ? (eeF - ebF) : WCAP;
      const int32_t* ids = (g.vcol + ebF) + (unsigned)lane;
#pragma unroll
      for (int k = 0; k < WCAP / 64; ++k)
        if (64 * k < ne) {
          if (lane + 64 * k < ne) DESCO_DMA4(ids + 64 * k, ecF + 64 * k);
        }
    }
    // id range of tile i + 3 (scalar loads, used in the next iteration)
    int64_t w0G;
    int nrG;
    DESCO_TILE_ROWS(tile + 3 * tstride, w0G, nrG)
    int ebG = 0, eeG = 0;
    if (S > 0 && nrG > 0) {
      const int32_t* q_ = rowptr_s + (g.row0 + w0G) * S;
      ebG = q_[0];
      eeG = q_[nrG * S];
    }
    // fused pooling: this tile's segment-end bitmap and first partial slot (wave-uniform address:
    // scalar loads, in flight under the whole tile)
    uint32_t pool_e = 0;
    int pool_s = 0;
    if constexpr (POOL) {
      const int t16 = __builtin_amdgcn_readfirstlane((int)(growC >> 4));
      pool_e = pool_bits_s[t16];
      pool_s = pool_slot_s[t16];
    }

    // ---- accumulator init: bias ----------------------------------------------------------------
    f32x4 q0, q1, q2, q3;                        // 16 rows x 64 columns: four 16-column tiles
    q0 = f32x4{bq0, bq0, bq0, bq0};
    q1 = f32x4{bq1, bq1, bq1, bq1};
    q2 = f32x4{bq2, bq2, bq2, bq2};
    q3 = f32x4{bq3, bq3, bq3, bq3};
    DESCO_BLOCK(0, G0)
    DESCO_BLOCK(1, G1)
    DESCO_BLOCK(2, G2)
    DESCO_BLOCK(3, G3)

    const int64_t grow_out = growC;
    const int nr_out = nrC;
    // ---- epilogue.  C/D map of a 16x16 tile: lane (c = lane&15, g = lane>>4) holds rows 4 g + e of
    //      column 16 t + c.  A 4x4 transpose over the lane quarters (tile t of quarter g <-> tile g of
    //      quarter t: v_permlane32_swap, then v_permlane16_swap) leaves lane = column with the 16 rows
    //      of the tile in registers, row 4 t + e in q_t[e]: every store is one full 256-byte row, and
    //      the pooling pass is a running sum in row order ------------------------------------------------
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      q0[e] = apply_act(q0[e], g.act, g.slope);
      q1[e] = apply_act(q1[e], g.act, g.slope);
      q2[e] = apply_act(q2[e], g.act, g.slope);
      q3[e] = apply_act(q3[e], g.act, g.slope);
    }
    {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // quarters {2,3} of q0 <-> quarters {0,1} of q2, likewise q1 / q3
        u32x2 t_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(q0[e]), __float_as_uint(q2[e]), false, false);
        q0[e] = __uint_as_float(t_[0]);
        q2[e] = __uint_as_float(t_[1]);
        t_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(q1[e]), __float_as_uint(q3[e]), false, false);
        q1[e] = __uint_as_float(t_[0]);
        q3[e] = __uint_as_float(t_[1]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // odd quarters of q0 <-> even quarters of q1, likewise q2 / q3
        u32x2 t_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(q0[e]), __float_as_uint(q1[e]), false, false);
        q0[e] = __uint_as_float(t_[0]);
        q1[e] = __uint_as_float(t_[1]);
        t_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(q2[e]), __float_as_uint(q3[e]), false, false);
        q2[e] = __uint_as_float(t_[0]);
        q3[e] = __uint_as_float(t_[1]);
      }
    }
    const int nru = __builtin_amdgcn_readfirstlane(nr_out);
#define DESCO_ROW(r_) ((r_) < 4 ? q0[(r_) & 3] : (r_) < 8 ? q1[(r_) & 3] : (r_) < 12 ? q2[(r_) & 3] : q3[(r_) & 3])
    if (!POOL || g.out) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < nru) {
          g.out[(grow_out + r) * g.ldo + lane] = DESCO_ROW(r);
          if (g.out2) g.out2[(grow_out - g.row0 + r) * g.ldo2 + lane] = DESCO_ROW(r);
        }
      }
    }
    if constexpr (POOL) {
      const uint32_t E = __builtin_amdgcn_readfirstlane(pool_e);
      int slot = __builtin_amdgcn_readfirstlane(pool_s);
      float* pp = g.pool_part + lane;
      float run = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < nru) {                                        // (wave-uniform)
          run += DESCO_ROW(r);
          if ((E >> r) & 1u) {                                // row r ends its segment (wave-uniform)
            pp[(int64_t)slot * 64] = run;
            ++slot;
            run = 0.f;
          }
        }
      }
      // the last segment of the tile continues in the next tile: its partial so far
      if (nru > 0 && !((E >> (nru - 1)) & 1u)) pp[(int64_t)slot * 64] = run;
    }
#undef DESCO_ROW
    // ---- rotate: next -> current, far -> next, the current buffers take tile i + 3's slices ---------
    if (nrN == 0) break;
    {
      int* t_ = rpC; rpC = rpN; rpN = rpF; rpF = t_;
      t_ = ecC; ecC = ecN; ecN = ecF; ecF = t_;
    }
    tile += tstride;
    w0C = w0N; nrC = nrN; growC = growN;
    w0N = w0F; nrN = nrF; growN = g.row0 + w0F;
    w0F = w0G; nrF = nrG;
    ebF = ebG; eeF = eeG;
    ebaseC = ebaseN;
    liveC = liveN;
  }
}

#undef DESCO_DMA4
#undef DESCO_ISSUE_SLOT
#undef DESCO_ISSUE_SELF
#undef DESCO_TAB_CUR
#undef DESCO_ISSUE_TAB
#undef DESCO_ISSUE_BLOCK
#undef DESCO_CUR
#undef DESCO_ISSUE2
#undef DESCO_CONSUME2
#undef DESCO_ZERO_SUMS
#undef DESCO_ANY_STAGED
#undef DESCO_CONSUME_TAB
#undef DESCO_COOP
#undef DESCO_FINISH
#undef DESCO_SLOT_LIVE
#undef DESCO_TILE_LIVE
#undef DESCO_PUT_F32
#undef DESCO_PUT_X6
#undef DESCO_M16
#undef DESCO_MFMA_HALF_X6
#undef DESCO_TAB_HALF
#undef DESCO_BLOCK
#undef DESCO_TILE_ROWS

template <int KB, int ST, bool LD64, bool POOL>
static void shmp16t_launch_one(const ShmpArgs& g, unsigned grid, hipStream_t st) {
  constexpr int WST = KB * 64 + 16;
  constexpr size_t w_floats = (size_t)3 * 64 * WST / 2;
  constexpr size_t shmem = sizeof(float) * (w_floats + (size_t)NW * WAVE_LDS);
  static_assert(shmem <= 160 * 1024, "SHMP layer (tile-ahead gathers): LDS budget exceeded");
  static DeviceOnce attr_once;        // function attributes are per device
  if (!attr_once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shmp_layer16t_kernel<KB, ST, LD64, POOL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.mark();
  }
  hipLaunchKernelGGL((shmp_layer16t_kernel<KB, ST, LD64, POOL>), dim3(grid), dim3(NW * 64), shmem, st, g,
                     g.vrowptr, g.pool_bits, g.pool_slot);
}

template <int KB>
static bool shmp16t_launch_st(const ShmpArgs& g, unsigned grid, hipStream_t st) {
  const bool ld64 = g.ldx == 64 && (g.st == 0 || g.ldy == 64 * g.st);
  if (g.pool_part) {
    if constexpr (KB == 3) {
      if (g.st != 2) return false;
      if (ld64)
        shmp16t_launch_one<3, 2, true, true>(g, grid, st);
      else
        shmp16t_launch_one<3, 2, false, true>(g, grid, st);
      return true;
    } else {
      return false;
    }
  }
#define DESCO_ONE(ST_)                                        \
  if (ld64)                                                   \
    shmp16t_launch_one<KB, ST_, true, false>(g, grid, st);    \
  else                                                        \
    shmp16t_launch_one<KB, ST_, false, false>(g, grid, st);
  switch (g.st) {
    case 0: DESCO_ONE(0) break;
    case 1: DESCO_ONE(1) break;
    case 2: DESCO_ONE(2) break;
    default: return false;
  }
#undef DESCO_ONE
  return true;
}

// x6 arguments validated by shmp_launch (shmp_layer.hip); g.wplanes set, g.sm <= 2
bool shmp16t_launch(const ShmpArgs& g, int cus, void* stream) {
  if (!g.wplanes || g.sm < 0 || g.sm > 2 || g.S > MAXS) return false;
  const int64_t ntiles = (g.num_rows + NW * WR - 1) / (NW * WR);
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  switch (g.sm) {
    case 0: return shmp16t_launch_st<1>(g, grid, (hipStream_t)stream);
    case 1: return shmp16t_launch_st<2>(g, grid, (hipStream_t)stream);
    default: return shmp16t_launch_st<3>(g, grid, (hipStream_t)stream);
  }
}

}  // namespace desco
