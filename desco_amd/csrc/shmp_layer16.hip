// Fused SHMP layer, 16-row wave tiles: the same layer as shmp_layer.hip (gnn_model.py:47-70 of the
// reference: per relation SAGEConv(aggr=add) neighbor sums, root/self Linear, HeteroConv sum, relu) in
// its bf16 6-product form, built for occupancy instead of tile size.
//
// Why a second tiling.  The 32-row kernel runs 8 waves per CU (2 per SIMD): its per-wave LDS (A image,
// row pointers, 512 staged source ids) fills the 160 KB beside the weight planes.  Its time follows
// T(w) = a + b / w in the waves per CU (8 -> 4 waves costs 1.44-1.53x): the
// gather latency is hidden by other waves, not by the two-source steps of one wave.  A wave tile of
// 16 rows (v_mfma_f32_16x16x32_bf16: four 16-column tiles of the 64 outputs) needs a quarter of the
// accumulator registers, half the A image and half the staged ids, so 16 waves (4 per SIMD, the
// 1024-thread block limit) fit beside three weight blocks -- if a wave stays within 128 registers:
//   * wave-uniform bookkeeping (tile, row range, CSR bases) is forced into SGPRs (readfirstlane);
//   * the next tile's row pointers and source ids go from global memory straight into LDS
//     (global_load_lds_dword) instead of waiting in registers until the tile switch.
// Measured (profiles/r2_h_ab_tile_rows.log): 8 x 32 rows -> 12 x 16 -> 16 x 16 waves x rows:
// Syn_1827 x4 34.0k -> 37.4k -> 39.6k graphs/s, MSRC+IMDB x8 253k -> 266k -> 281k, COX2 x64 664k ->
// 681k -> 696k.
//
// Everything else is the 32-row kernel's design (see its header): persistent blocks, wave-autonomous
// tiles in XCD-aware order, CSR row pointers double-buffered in LDS, the first two sources of every
// row in flight under the previous block's MFMAs, cooperative path for heavy rows, table pseudo block
// added in the C/D layout, optional fused pooling.  Differences:
//   * lane group g8 gathers rows g8 and 8 + g8 (it = 0, 1) instead of four rows;
//   * the A planes are [3][16][32] bf16 without padding (XOR-swizzled 16-byte chunks);
//   * the epilogue transposes the four C/D tiles across the lane quarters (v_permlane32_swap +
//     v_permlane16_swap) so that lane = column: stores are full 256-byte rows, the pooling pass a
//     running sum in row order with wave-uniform control flow (16-row tiles: pool index of 16 rows).
#include "common_device.hpp"
#include "shmp_args.hpp"

namespace desco {

constexpr int WR = 16;        // rows per wave
constexpr int AH = 36;        // half-K fp32 table image row stride (floats): conflict-free ds_read_b32 / ds_write_b128
constexpr int APS = 32;       // half-K bf16 plane row stride (shorts, 64 B), chunks XOR-swizzled
constexpr int MAXS = 4;       // relation slots stored per row
constexpr int RPN = WR * MAXS + 2;
// (round 6 sweep on the f16 form with 12 waves, same-box A/B, count-row launch: 5 steps +2.5 %; 14 / 20 / 32 steps -0.9 %
//  on Syn_1827 shapes and -1.4 % on MSRC-21 + IMDB shapes, COX2 shapes unchanged -- NOT taken: a row of 21..42 sources
//  is then summed two at a time instead of by the cooperative path's eight partial sums, and which rows of a tile lie
//  beyond the WCAP staged ids depends on the tile, so the 2-rank-vs-1-rank deviation of the un-chunked pipeline rose
//  from 5e-5 to 1.05e-4 of 1 + |count|, past its 1e-4 gate (tests/test_multirank_gpu.py))
#ifndef SHMP16_EXTRA_STEPS
#define SHMP16_EXTRA_STEPS 9
#endif
#ifndef SHMP16_WCAP
#define SHMP16_WCAP 200
#endif
constexpr int EXTRA_STEPS = SHMP16_EXTRA_STEPS; // batched 2-source steps after the prefetched one (<= 20 sources per row)
constexpr int WCAP = SHMP16_WCAP;     // source ids staged per wave and buffer (longer slices fall back to global)
constexpr int A_FLOATS = 3 * WR * APS / 2;               // A region per wave: max(16*36, 3*16*32/2) floats (bf16x6 form)
constexpr int WAVE_LDS = A_FLOATS + 2 * RPN + 2 * WCAP;  // floats per wave
static_assert(A_FLOATS >= WR * AH, "the fp32 image must fit in the plane region");
// fp16 three-product form: two planes (512 floats) < the fp32 table image (576); + 16 row scales per wave
constexpr int A_FLOATS_F16 = WR * AH;
constexpr int WAVE_LDS_F16 = A_FLOATS_F16 + 2 * RPN + 2 * WCAP + WR;
static_assert(A_FLOATS_F16 >= 2 * WR * APS / 2, "the fp16 planes must fit in the image region");

// absent sources of a batched gather step read this row instead of being predicated away
__device__ __attribute__((aligned(16))) float shmp16_zero_row[64] = {};

using bf16x8 = __attribute__((ext_vector_type(8))) short;

__device__ __forceinline__ void f4add(float4& a, const float4 b) {
#if !defined(SHMP16_F4ADD_PACKED)
  // four plain adds: left to itself hipcc pairs them into two v_pk_add_f32, which cost more issue time than the four
  // (tu_no_packed_f32_begin.hpp; the whole file without packed selection spills).  Same-box A/B
  // (profiles/r6_aq_ab_f4add_scalar.log): count-row launch 6.45 -> 6.32 ms on Syn_1827 shapes (many gather steps per
  // tile), 1.206 -> 1.208 on COX2 shapes
  asm("v_add_f32 %0, %0, %1" : "+v"(a.x) : "v"(b.x));
  asm("v_add_f32 %0, %0, %1" : "+v"(a.y) : "v"(b.y));
  asm("v_add_f32 %0, %0, %1" : "+v"(a.z) : "v"(b.z));
  asm("v_add_f32 %0, %0, %1" : "+v"(a.w) : "v"(b.w));
#else
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
#endif
}

// 4 bytes per lane from global straight into LDS: lane i's dword lands at dst_[i] (dst_ wave-uniform).
// Completion is counted by vmcnt like any vector load, but the compiler does not order later LDS reads
// behind it: the kernel waits explicitly where it first reads the data.
#define DESCO_DMA4(src_, dst_)                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_),      \
                                   (__attribute__((address_space(3))) void*)(dst_), 4, 0, 0)
// ---- gather machinery (macros: every temporary is a named register) ----------------
// They use the enclosing scope's rp, ec, ebase, grow0, nr, xb, yb, zrow, g, S, g8, l8 and the
// registers lo*/hi* (sums), u*/w* (loads in flight), c*/n* (cursors).
#define DESCO_CUR(it_, slot_)                              \
  {                                                        \
    const int v_ = ((it_) * 8 + g8) * S + (slot_);         \
    c##it_ = rp[v_] - ebase;                               \
    n##it_ = rp[v_ + 1] - ebase;                           \
  }
#define DESCO_CURS(slot_) DESCO_CUR(0, slot_) DESCO_CUR(1, slot_)
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
    lo1 = lo0;                                              \
    hi0 = lo0; hi1 = lo0;                                   \
  }
#define DESCO_ANY_STAGED()                                                                    \
  __any((c0 < (n0 < WCAP ? n0 : WCAP)) | (c1 < (n1 < WCAP ? n1 : WCAP)))
// the row itself: rows beyond nr re-read the wave's last valid row (never stored)
// (round 6, SHMP16_SELF_EARLY: into registers of their own -- s -- at the tile top, next to the first relation step)
#define DESCO_ISSUE_SELF(it_)                                                                  \
  {                                                                                            \
    const int r_ = (it_) * 8 + g8;                                                             \
    if constexpr (SELFDEG) {                                                                   \
      /* the row from its own slot degrees (row pointers of the tile: LDS), degree_affine's arithmetic */ \
      const int vb_ = (r_ < nr ? r_ : nr - 1) * S;                                             \
      const float* cq_ = cself + 4 * l8;                                                       \
      float4 a0_ = *reinterpret_cast<const float4*>(cq_ + S * 64);                             \
      float4 a1_ = *reinterpret_cast<const float4*>(cq_ + S * 64 + 32);                        \
      _Pragma("unroll") for (int s_ = 0; s_ < MAXS; ++s_) {                                    \
        if (s_ < S) {                                                                          \
          const float dd_ = (float)(rp[vb_ + s_ + 1] - rp[vb_ + s_]);                          \
          const float4 q0_ = *reinterpret_cast<const float4*>(cq_ + s_ * 64);                  \
          const float4 q1_ = *reinterpret_cast<const float4*>(cq_ + s_ * 64 + 32);             \
          a0_.x = fmaf(dd_, q0_.x, a0_.x); a0_.y = fmaf(dd_, q0_.y, a0_.y);                    \
          a0_.z = fmaf(dd_, q0_.z, a0_.z); a0_.w = fmaf(dd_, q0_.w, a0_.w);                    \
          a1_.x = fmaf(dd_, q1_.x, a1_.x); a1_.y = fmaf(dd_, q1_.y, a1_.y);                    \
          a1_.z = fmaf(dd_, q1_.z, a1_.z); a1_.w = fmaf(dd_, q1_.w, a1_.w);                    \
        }                                                                                      \
      }                                                                                        \
      a0_.x = fmaxf(a0_.x, 0.f); a0_.y = fmaxf(a0_.y, 0.f); a0_.z = fmaxf(a0_.z, 0.f); a0_.w = fmaxf(a0_.w, 0.f); \
      a1_.x = fmaxf(a1_.x, 0.f); a1_.y = fmaxf(a1_.y, 0.f); a1_.z = fmaxf(a1_.z, 0.f); a1_.w = fmaxf(a1_.w, 0.f); \
      DESCO_SELF_REG(it_, 0) = a0_;                                                            \
      DESCO_SELF_REG(it_, 1) = a1_;                                                            \
    } else {                                                                                   \
      const float* p_ = xsb + (grow0 + (r_ < nr ? r_ : nr - 1)) * LDXS;                      \
      DESCO_SELF_REG(it_, 0) = *reinterpret_cast<const float4*>(p_);                           \
      DESCO_SELF_REG(it_, 1) = *reinterpret_cast<const float4*>(p_ + 32);                      \
    }                                                                                          \
  }
// table pseudo block: the first source of table slot 0 (-> u) and of table slot 1 (-> w) of row it_
#define DESCO_TAB_CUR(it_)                                                                  \
  const int v_ = ((it_) * 8 + g8) * S + g.sm;                                               \
  const int ca_ = rp[v_] - ebase, na_ = rp[v_ + 1] - ebase;                                 \
  const int nb_ = ST > 1 ? rp[v_ + 2] - ebase : na_;                                        \
  const bool k0_ = ca_ < (na_ < WCAP ? na_ : WCAP);                                         \
  const bool k1_ = ST > 1 && na_ < (nb_ < WCAP ? nb_ : WCAP);
// (a table slot without a source among the tile's rows -- live bits 8, 9 -- is neither loaded nor added:
// only the rows next to the canonical node have one, half of the Syn_1827 tiles have none at all)
#define DESCO_ISSUE_TAB(it_)                                                                \
  {                                                                                         \
    DESCO_TAB_CUR(it_)                                                                      \
    const int i0_ = ec[k0_ ? ca_ : 0], i1_ = ec[k1_ ? na_ : 0];                             \
    if (live & 0x100) {                                                                     \
      const float* p0_ = k0_ ? yb + (int64_t)i0_ * LDY : zrow;                              \
      u##it_##0 = *reinterpret_cast<const float4*>(p0_);                                    \
      u##it_##1 = *reinterpret_cast<const float4*>(p0_ + 32);                               \
    }                                                                                       \
    if (ST > 1 && (live & 0x200)) {                                                         \
      const float* p1_ = k1_ ? yb + 64 + (int64_t)i1_ * LDY : zrow;                         \
      w##it_##0 = *reinterpret_cast<const float4*>(p1_);                                    \
      w##it_##1 = *reinterpret_cast<const float4*>(p1_ + 32);                               \
    }                                                                                       \
  }
// consume the step; leave the cursor of table slot 0 in (c, n) and of slot 1 in (d, m)
#define DESCO_CONSUME_TAB(it_)                                                              \
  {                                                                                         \
    DESCO_TAB_CUR(it_)                                                                      \
    if (live & 0x100) {                                                                     \
      f4add(lo##it_, u##it_##0);                                                            \
      f4add(hi##it_, u##it_##1);                                                            \
    }                                                                                       \
    if (ST > 1 && (live & 0x200)) {                                                         \
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
// finish a gathered block whose first step is already in flight: consume it, up to EXTRA_STEPS more
// batched steps (two sources per row each: all four rows of a lane group advance together), then
// the cooperative path for rows that are heavier still (one row at a time, the whole wave on it)
#define DESCO_FINISH(base_, ld_)                                                           \
  {                                                                                        \
    DESCO_CONSUME2(0) DESCO_CONSUME2(1)                                                    \
    for (int st_ = 0; st_ < EXTRA_STEPS && DESCO_ANY_STAGED(); ++st_) {                    \
      DESCO_ISSUE2(0, base_, ld_) DESCO_ISSUE2(1, base_, ld_)                              \
      DESCO_CONSUME2(0) DESCO_CONSUME2(1)                                                  \
    }                                                                                      \
    if (__any((c0 < n0) | (c1 < n1))) {                                                    \
      DESCO_COOP(0, base_, ld_) DESCO_COOP(1, base_, ld_)                                  \
    }                                                                                      \
  }
// first step of block b_ (cursors + loads); nothing waits on the loads here
#define DESCO_ISSUE_BLOCK(b_)                                                              \
  {                                                                                        \
    if ((b_) < KB - 1) {                                                                   \
      DESCO_CURS(b_)                                                                       \
      DESCO_ISSUE2(0, xb, LDX) DESCO_ISSUE2(1, xb, LDX)                                \
    } else if ((b_) == KB - 1) {                                                           \
      DESCO_ISSUE_SELF(0) DESCO_ISSUE_SELF(1)                                              \
    } else {                                                                               \
      DESCO_ISSUE_TAB(0) DESCO_ISSUE_TAB(1)                                                \
    }                                                                                      \
  }
// bit s of `live`: relation slot s (an MFMA slot) has at least one source among the wave's 16 rows
#define DESCO_SLOT_ANY(s_)                                                                 \
  (__any((rp[(0 * 8 + g8) * S + (s_) + 1] > rp[(0 * 8 + g8) * S + (s_)]) |                  \
         (rp[(1 * 8 + g8) * S + (s_) + 1] > rp[(1 * 8 + g8) * S + (s_)])) != 0)
#define DESCO_SLOT_LIVE(s_) (DESCO_SLOT_ANY(s_) ? 1 << (s_) : 0)
#define DESCO_TILE_LIVE()                                        \
  {                                                              \
    live = 0;                                                    \
    if (KB - 1 > 0) live |= DESCO_SLOT_LIVE(0);                  \
    if (KB - 1 > 1) live |= DESCO_SLOT_LIVE(1);                  \
    if (KB - 1 > 2) live |= DESCO_SLOT_LIVE(2);                  \
    if (ST > 0) live |= DESCO_SLOT_ANY(g.sm) ? 0x100 : 0;        \
    if (ST > 1) live |= DESCO_SLOT_ANY(g.sm + 1) ? 0x200 : 0;    \
  }
// first step of the first LIVE block after block a_ (a_ = -1: of the tile); dead slots are left
// out of the software pipeline altogether, so the block behind one is not issued late.  The slot
// index is a wave-uniform runtime value here (one copy of the gather issue code per site).
#ifndef SHMP16_SELF_LATE
#define SHMP16_SELF_EARLY 1
#endif
#ifdef SHMP16_SELF_EARLY
// The rows themselves (the self block's operand) need no index: they are requested at the tile top, together with the
// first step of the first live relation block, into registers of their own -- one exposed round trip per tile less
// (the self rows used to be requested under the MFMAs of the last relation block, which are 200 cycles long); the table
// step moves up with them: it goes out behind the LAST relation block instead of behind the self block.
#define DESCO_SELF_REG(it_, h_) s##it_##h_
#define DESCO_ISSUE_AFTER(a_)                                                              \
  {                                                                                        \
    if ((a_) < KB - 1) {                                                                   \
      int nb_ = KB - 1;                                                                    \
      if ((a_) + 3 < KB - 1 && ((live >> ((a_) + 3)) & 1)) nb_ = (a_) + 3;                  \
      if ((a_) + 2 < KB - 1 && ((live >> ((a_) + 2)) & 1)) nb_ = (a_) + 2;                  \
      if ((a_) + 1 < KB - 1 && ((live >> ((a_) + 1)) & 1)) nb_ = (a_) + 1;                  \
      if (nb_ < KB - 1) {                                                                  \
        DESCO_CURS(nb_)                                                                    \
        DESCO_ISSUE2(0, xb, LDX) DESCO_ISSUE2(1, xb, LDX)                                  \
      } else if (ST > 0 && (live & 0x300)) {                                               \
        DESCO_ISSUE_TAB(0) DESCO_ISSUE_TAB(1)                                              \
      }                                                                                    \
      if ((a_) < 0) {                                                                      \
        DESCO_ISSUE_SELF(0) DESCO_ISSUE_SELF(1)                                            \
      }                                                                                    \
    }                                                                                      \
  }
#else
#define DESCO_SELF_REG(it_, h_) u##it_##h_
#define DESCO_ISSUE_AFTER(a_)                                                              \
  {                                                                                        \
    if ((a_) < KB - 1) {                                                                   \
      int nb_ = KB - 1;                                                                    \
      if ((a_) + 3 < KB - 1 && ((live >> ((a_) + 3)) & 1)) nb_ = (a_) + 3;                  \
      if ((a_) + 2 < KB - 1 && ((live >> ((a_) + 2)) & 1)) nb_ = (a_) + 2;                  \
      if ((a_) + 1 < KB - 1 && ((live >> ((a_) + 1)) & 1)) nb_ = (a_) + 1;                  \
      if (nb_ < KB - 1) {                                                                  \
        DESCO_CURS(nb_)                                                                    \
        DESCO_ISSUE2(0, xb, LDX) DESCO_ISSUE2(1, xb, LDX)                                  \
      } else {                                                                             \
        DESCO_ISSUE_SELF(0) DESCO_ISSUE_SELF(1)                                            \
      }                                                                                    \
    } else if ((a_) + 1 < NB && (live & 0x300)) {                                          \
      DESCO_ISSUE_BLOCK((a_) + 1)                                                          \
    }                                                                                      \
  }
#endif
// write one fp32 half image (every lane writes: row = it*8 + g8, 4 floats at 4*l8)
#define DESCO_PUT_F32(av_, it_)                         \
  {                                                     \
    *reinterpret_cast<float4*>(Aw + ((it_) * 8 + g8) * AH + 4 * l8) = av_;   \
  }
// write one half image as three bf16 planes (row = it*8 + g8, 4 bf16 at 4*l8 of every plane).  Plane
// rows are 64 B with no padding: the 16-byte chunk k/8 of row r sits at chunk (k/8) ^ (-(r/4) & 3), which
// makes both this write (ds_write_b64) and the fragment read conflict-free (ds_read_b128 is served in
// the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: tools/micro/lds_banks.py)
#define DESCO_SPLIT(a_, b_, h_, m_, l_) split2_bf16x3(a_, b_, h_, m_, l_)
#define DESCO_PUT_X6(av_, it_)                                                  \
  {                                                                             \
    uint32_t h0_, m0_, l0_, h1_, m1_, l1_;                                      \
    DESCO_SPLIT(av_.x, av_.y, h0_, m0_, l0_);                                     \
    DESCO_SPLIT(av_.z, av_.w, h1_, m1_, l1_);                                     \
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
// ---- fp16 three-product form (common_device.hpp: x s = hi + lo, hi*hi + hi*lo + lo*hi) --------------------------------
// Row scales.  The operands of a row are multiplied by ONE power of two per row, re-chosen only when a K block does not
// fit under it.  The lane group that holds a row reduces the maximum of its 64 gathered sums (3 DPP steps over the
// group's 8 lanes); the accumulators (C/D layout: lane quarter g holds rows 4 g + e) follow a change of scale by an
// exact multiplication and leave the scales in the epilogue.
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
// (a chain: hipcc folds it into four v_max3_f32 with |.| source modifiers)
#define DESCO_ABSMAX8(a_, b_)                                                                                   \
  fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(fabsf(a_.x), fabsf(a_.y)), fabsf(a_.z)), fabsf(a_.w)), fabsf(b_.x)), \
                    fabsf(b_.y)), fabsf(b_.z)), fabsf(b_.w))
// max over the 8 lanes of a lane group (non-negative floats compared as unsigned integers)
#define DESCO_GROUP_MAX(m_)                                                                                     \
  {                                                                                                             \
    uint32_t v_ = __float_as_uint(m_);                                                                          \
    uint32_t o_ = __builtin_amdgcn_update_dpp(0u, v_, 0xB1, 0xf, 0xf, true);  /* quad_perm [1,0,3,2] */         \
    v_ = v_ > o_ ? v_ : o_;                                                                                     \
    o_ = __builtin_amdgcn_update_dpp(0u, v_, 0x4E, 0xf, 0xf, true);           /* quad_perm [2,3,0,1] */         \
    v_ = v_ > o_ ? v_ : o_;                                                                                     \
    o_ = __builtin_amdgcn_update_dpp(0u, v_, 0x141, 0xf, 0xf, true);          /* row_half_mirror */             \
    v_ = v_ > o_ ? v_ : o_;                                                                                     \
    m_ = __uint_as_float(v_);                                                                                   \
  }
// Scales of this lane group's two rows for the block whose sums are complete.  A row keeps its scale from block to
// block as long as the new block fits under it (scaled maximum < 2^15.9; a fresh scale puts the maximum at 2^12 ..
// 2^13, so later blocks may be 8x larger; blocks that are smaller lose absolute precision only relative to the row's
// LARGEST block, which is what their common output sum is accurate to anyway): the common case is a maximum, two
// multiplications and one wave vote.  When a row of the tile does not fit (or has no scale yet: the tile's first
// block with a source for it), the rows that need it take a new scale, the scales cross to the C/D layout through LDS
// (sentinel 0 = unchanged) and the accumulators follow by an exact multiplication.
#define DESCO_BLOCK_SCALES()                                                                                    \
  {                                                                                                             \
    float m0_ = DESCO_ABSMAX8(lo0, hi0), m1_ = DESCO_ABSMAX8(lo1, hi1);                                         \
    DESCO_GROUP_MAX(m0_) DESCO_GROUP_MAX(m1_)                                                                   \
    const bool o0_ = m0_ * sc0 > 60000.f || (sc0 == 0.f && m0_ > 0.f);                                          \
    const bool o1_ = m1_ * sc1 > 60000.f || (sc1 == 0.f && m1_ > 0.f);                                          \
    if (__any(o0_ | o1_)) {                                                                                     \
      const float s0_ = f16_scale_for(m0_) * 0.25f, s1_ = f16_scale_for(m1_) * 0.25f;                           \
      if (o0_) sc0 = s0_;                                                                                       \
      if (o1_) sc1 = s1_;                                                                                       \
      rs[g8] = o0_ ? s0_ : 0.f;                                                                                 \
      rs[8 + g8] = o1_ ? s1_ : 0.f;                                                                             \
      const float4 n_ = *reinterpret_cast<const float4*>(rs + 4 * (lane >> 4));                                 \
      const float n0_ = n_.x > 0.f ? n_.x : cs[0], n1_ = n_.y > 0.f ? n_.y : cs[1];                             \
      const float n2_ = n_.z > 0.f ? n_.z : cs[2], n3_ = n_.w > 0.f ? n_.w : cs[3];                             \
      const f32x4 r_ = {n0_ * pow2_inverse(cs[0]), n1_ * pow2_inverse(cs[1]), n2_ * pow2_inverse(cs[2]),        \
                        n3_ * pow2_inverse(cs[3])};                                                             \
      q0 *= r_; q1 *= r_; q2 *= r_; q3 *= r_;                                                                   \
      cs = f32x4{n0_, n1_, n2_, n3_};                                                                           \
    }                                                                                                           \
  }
// write one half image as two fp16 planes (same addressing as the bf16 planes)
#define DESCO_PUT_F16(av_, it_, sc_)                                            \
  {                                                                             \
    uint32_t h0_, l0_, h1_, l1_;                                                \
    split2_f16x2(av_.x * (sc_), av_.y * (sc_), h0_, l0_);                       \
    split2_f16x2(av_.z * (sc_), av_.w * (sc_), h1_, l1_);                       \
    short* d_ = Ap + ((it_) * 8 + g8) * APS +                                   \
                ((((l8 >> 1) ^ (0 - ((it_) * 2 + (g8 >> 2)))) & 3) << 3) + 4 * (l8 & 1); \
    *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                       \
    *reinterpret_cast<uint2*>(d_ + WR * APS) = make_uint2(l0_, l1_);            \
  }
#define DESCO_F16(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0);
// 12 fp16 MFMAs (3-product split) on the staged half (32 k) of block b_
#define DESCO_MFMA_HALF_F16(b_, h_)                                                               \
  {                                                                                               \
    const short* ap_ = Ap + (lane & 15) * APS + ((((lane >> 4) ^ (0 - (lane >> 2))) & 3) << 3);   \
    const short* bp_ = Wp + (lane & 15) * WST + (b_) * 64 + (h_) * 32 + 8 * (lane >> 4);          \
    const f16x8 ah_ = *reinterpret_cast<const f16x8*>(ap_);                                       \
    const f16x8 al_ = *reinterpret_cast<const f16x8*>(ap_ + WR * APS);                            \
    const f16x8 b0h_ = *reinterpret_cast<const f16x8*>(bp_);                                      \
    const f16x8 b0l_ = *reinterpret_cast<const f16x8*>(bp_ + WPL);                                \
    const f16x8 b1h_ = *reinterpret_cast<const f16x8*>(bp_ + 16 * WST);                           \
    const f16x8 b1l_ = *reinterpret_cast<const f16x8*>(bp_ + 16 * WST + WPL);                     \
    DESCO_F16(al_, b0h_, q0) DESCO_F16(al_, b1h_, q1)                                             \
    DESCO_F16(ah_, b0l_, q0) DESCO_F16(ah_, b1l_, q1)                                             \
    const f16x8 b2h_ = *reinterpret_cast<const f16x8*>(bp_ + 32 * WST);                           \
    const f16x8 b2l_ = *reinterpret_cast<const f16x8*>(bp_ + 32 * WST + WPL);                     \
    const f16x8 b3h_ = *reinterpret_cast<const f16x8*>(bp_ + 48 * WST);                           \
    const f16x8 b3l_ = *reinterpret_cast<const f16x8*>(bp_ + 48 * WST + WPL);                     \
    DESCO_F16(ah_, b0h_, q0) DESCO_F16(ah_, b1h_, q1)                                             \
    DESCO_F16(al_, b2h_, q2) DESCO_F16(al_, b3h_, q3)                                             \
    DESCO_F16(ah_, b2l_, q2) DESCO_F16(ah_, b3l_, q3)                                             \
    DESCO_F16(ah_, b2h_, q2) DESCO_F16(ah_, b3h_, q3)                                             \
  }
// table half rows in the scaled accumulators: the (unscaled, fp32) table values enter at the rows' current scales
// (the accumulators are in units of row scale x weight scale)
#define DESCO_TAB_HALF_F16(qa_, qb_)                                                        \
  {                                                                                        \
    const f32x4 cw_ = cs * wsc;                                                            \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                     \
      const float* t_ = Aw + (4 * (lane >> 4) + e_) * AH + (lane & 15);                    \
      qa_[e_] += t_[0] * cw_[e_];                                                          \
      qb_[e_] += t_[16] * cw_[e_];                                                         \
    }                                                                                      \
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

// KB = sm + 1 resident weight blocks (1..3), ST table slots (0..2),
// LD64: x and out rows are 64 floats and ytab rows 64*ST floats apart (the product path's layouts):
// source-row addresses then need a shift instead of a 64-bit multiply per gathered row, and the stores
// of a tile are one address with immediate offsets
// POOL: fused pooling epilogue (instantiated for the count-row launches only)
template <int NW, int KB, int ST, bool LD64, bool POOL, bool F16, bool SELFDEG = false>
__global__ __launch_bounds__(NW * 64) void shmp_layer16_kernel(ShmpArgs g, const int32_t* __restrict__ rowptr_s,
                                                              const uint32_t* __restrict__ pool_bits_s,
                                                              const int32_t* __restrict__ pool_slot_s) {
  // rowptr_s / pool_*_s = g.vrowptr / g.pool_bits / g.pool_slot once more, as read-only restrict
  // parameters: their wave-uniform loads then go through the scalar cache (s_load, counted by lgkmcnt)
  // instead of queueing behind the gathers in the in-order vector memory pipe
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WST = KB * 64 + 16;                        // weight plane row stride (shorts): 32 B of padding, conflict-free B fragments
  constexpr int WPL = 64 * WST;                            // shorts per weight plane
  constexpr int NP = F16 ? 2 : 3;                          // operand planes (fp16 hi / lo; bf16 hi / mid / lo)
  constexpr int W_FLOATS = NP * WPL / 2;
  constexpr int WAVE_LDS_ = F16 ? WAVE_LDS_F16 : WAVE_LDS;
  constexpr int A_FLOATS_ = F16 ? A_FLOATS_F16 : A_FLOATS;
  short* Wp = reinterpret_cast<short*>(lds);               // [NP][64 n][WST]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile bookkeeping stays in SGPRs
  float* Aw = lds + W_FLOATS + wave * WAVE_LDS_;           // fp32 half image [16][36] (table block)
  short* Ap = reinterpret_cast<short*>(Aw);                // operand planes [NP][16][32] of a half image
  int* rpb = reinterpret_cast<int*>(Aw + A_FLOATS_);       // 2 x [16*S+1] row pointers (absolute)
  int* ecb = rpb + 2 * RPN;                                // 2 x [WCAP] source ids (current / next tile)
  int* ec = ecb;
  float* rs = reinterpret_cast<float*>(ecb + 2 * WCAP);    // F16: [16] row scales of the block being staged
  (void)rs;
  float* biasL = lds + W_FLOATS + NW * WAVE_LDS_;          // [64] bias (zeros without one), block-shared
  int* next_sub = reinterpret_cast<int*>(biasL + 64);      // the block's tile hand-out counter
  // SELFDEG (its own instantiation: the launch's own rows recomputed from their slot degrees, see shmp_args.hpp): the
  // [S + 1][64] coefficients, block-shared
  float* cself = biasL + 64 + 4;
  (void)cself;

  // ---- resident weights -------------------------------------------------------------------
  {
    // global planes [NP][64][KB*64] -> LDS [NP][64][WST], 16 bytes at a time
    constexpr int CH = KB * 8;                             // uint4 chunks per row
    for (int i = tid; i < NP * 64 * CH; i += NW * 64) {
      const int row = i / CH, ch = i - row * CH;           // row = plane*64 + n
      *reinterpret_cast<uint4*>(Wp + row * WST + 8 * ch) =
          *reinterpret_cast<const uint4*>(g.wplanes + (int64_t)row * (KB * 64) + 8 * ch);
    }
  }
  if (tid < 64) biasL[tid] = g.bias ? g.bias[tid] : 0.f;
  if constexpr (SELFDEG)
    for (int i = tid; i < (g.S + 1) * 64; i += NW * 64) cself[i] = g.self_coef[i];
  if (tid == 0) *next_sub = 0;
  __syncthreads();

  const int g8 = lane >> 3, l8 = lane & 7;                 // 8 groups of 8 lanes: one half row each
  const float wsc = F16 ? g.wscale[0] : 1.f;               // power-of-two scale of the weight planes ...
  const float winv = F16 ? g.wscale[1] : 1.f;              // ... and its inverse
  (void)wsc;
  const int S = g.S;
  const int nslot = WR * S + 1;                            // <= 65: at most 2 per lane
  const int64_t ntiles = (g.num_rows + NW * WR - 1) / (NW * WR);
  constexpr int NB = KB + (ST > 0 ? 1 : 0);                // K blocks incl. the table pseudo block
  const int64_t LDX = LD64 ? 64 : g.ldx, LDY = LD64 ? 64 * (ST > 0 ? ST : 1) : g.ldy;
  const int64_t LDO = LD64 ? 64 : g.ldo;
  const float* xb = g.x + 4 * l8;
  const float* xsb = g.xself ? g.xself + 4 * l8 : xb;      // the launch's own rows (self block): x, or another tensor
  const int64_t LDXS = g.xself ? g.ldxs : LDX;
  const float* yb = ST > 0 ? g.ytab + 4 * l8 - g.ytab_row0 * LDY : nullptr;
  const float* zrow = shmp16_zero_row + 4 * l8;
  (void)yb;

  // ---- this wave's first tile ---------------------------------------------------------------
  // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch) and its 4 MB L2,
  // so a neighborhood's rows -- the sources of all its tiles -- should be gathered by ONE XCD.  Measured
  // (profiles/r2_f_ab_xcd_order.log): +0.5 % on Syn_1827 / MSRC+IMDB shapes, 0 on COX2 shapes (the
  // gathers are not what bounds the kernel); contiguous eighths per XCD were 8 % SLOWER on Syn shapes
  // (the dataset is ordered by graph size: the XCD with the dense end finishes last).  Speed only: any
  // block -> XCD placement gives the same result.
  int64_t tile, tend = ntiles;
  int tstride = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    // chunks of (grid / 8) consecutive tiles go round robin over the XCDs: XCD x works on the 32
    // neighbouring tiles of chunk 8 j + x in sweep j (locality), heavy and light regions of the dataset
    // are spread over all XCDs (balance)
    tile = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  } else {
    tile = blockIdx.x;
  }
  // A block's 16-row wave tiles are handed out IN ORDER from a counter in LDS (sub-tile i = wave tile
  // i % NW of the block's (i / NW)-th block tile): whatever the waves' speeds, the tiles in flight in a
  // block are consecutive -- a window of a few hundred rows that also holds most of their sources (a
  // neighborhood's rows are contiguous).  With a fixed wave -> tile map the waves drift apart by whole
  // sweeps and the L2 saw 3.3 row fetches per row on Syn_1827 shapes (hit rate 36 %).  A wave keeps two
  // indices: the tile it works on and the next one (id range, CSR slice and ids in flight under this tile); a third --
  // the id range a tile earlier still, rounds 2-3 -- widened the span of rows the block's waves work on at a time
  // (round 4: -1 % time on the dense shapes without it, +7 % with a fourth).
#define DESCO_NEXT_SUB() __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(next_sub, 1) : 0)
#define DESCO_SUB_ROWS(i_, w0_, ok_)                                                          \
  {                                                                                           \
    const int64_t bt_ = tile + (int64_t)((i_) / NW) * tstride;                                \
    w0_ = bt_ * (NW * WR) + ((i_) % NW) * WR;                                                 \
    ok_ = bt_ < tend && w0_ < g.num_rows;                                                     \
  }
  int sub_n1;
  int64_t w0;
  {
    const int i0_ = DESCO_NEXT_SUB();
    sub_n1 = DESCO_NEXT_SUB();
    bool ok_;
    DESCO_SUB_ROWS(i0_, w0, ok_)
    if (!ok_) return;                                      // no barrier below: idle waves may leave
  }
  int nr = (int)((g.num_rows - w0) < WR ? (g.num_rows - w0) : WR);
  int64_t grow0 = g.row0 + w0;
  int cur = 0;
  int* rp = rpb;
  if (S > 0) {                                             // S == 0: no CSR at all (plain row-wise Linear)
    const int nptr = nr * S + 1;
    for (int i = lane; i < nslot; i += 64)
      rp[i] = g.vrowptr[grow0 * S + (i < nptr ? i : nptr - 1)];
    const int eb = __builtin_amdgcn_readfirstlane(rp[0]);
    const int ecnt = __builtin_amdgcn_readfirstlane(rp[WR * S]) - eb;
    for (int i = lane; i < ecnt && i < WCAP; i += 64) ec[i] = g.vcol[eb + i];
  } else if (lane == 0) {
    rp[0] = 0;
    rpb[RPN] = 0;
  }
  int ebase = __builtin_amdgcn_readfirstlane(rp[0]);     // (wave-uniform values are kept in SGPRs)

  float4 lo0, lo1, hi0, hi1;                               // gathered sums of the current block
  float4 u00, u01, u10, u11;                               // in flight: first source (lo, hi) of row it
  float4 w00, w01, w10, w11;                               // in flight: second source
  w00 = w01 = w10 = w11 = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef SHMP16_SELF_EARLY
  float4 s00, s01, s10, s11;                               // in flight: the rows themselves (requested at the tile top)
#endif
  int c0 = 0, c1 = 0, n0 = 0, n1 = 0;                      // cursors [c, n) rel. to ebase
  // bit b: relation slot b has at least one source among this wave's 16 rows.  A slot that is empty
  // for the whole wave tile (triangle edges in molecule graphs, tride edges in clique unions) is an
  // all-zero K block: it is left out of the tile's block sequence (wave-uniform; it would add exact
  // zeros), and the first gather step of the block behind it is issued in its place
  int live = 0;
  DESCO_TILE_LIVE()
  DESCO_ISSUE_AFTER(-1)

  for (;;) {
    int* rpn = rpb + (cur ^ 1) * RPN;
    // ---- prefetch the row pointers of the next tile (registers now, LDS later) --------------
    int64_t w0n;
    bool has_next;
    DESCO_SUB_ROWS(sub_n1, w0n, has_next)
    const int nrn = has_next ? (int)((g.num_rows - w0n) < WR ? (g.num_rows - w0n) : WR) : 0;
    int* ecn = ecb + (cur ^ 1) * WCAP;
    // id range [ebn_cur, een_cur) of the next tile: two scalar loads, consumed behind this tile's last relation-slot
    // block (the LDS-direct load of the ids).  A wave holds TWO tiles, this one and the next: a third (its id range a
    // tile earlier still) widens the span of rows the workgroup's waves work on at a time, and with it the L2 misses.
    int ebn_cur = 0, een_cur = 0;
    if (S > 0 && has_next) {
      const int32_t* q_ = rowptr_s + (g.row0 + w0n) * S;
      ebn_cur = q_[0];
      een_cur = q_[nrn * S];
    }
    // fused pooling: this tile's segment-end bitmap and first partial slot (wave-uniform address:
    // scalar loads, in flight under the whole tile)
    uint32_t pool_e = 0;
    int pool_s = 0;
    if constexpr (POOL) {
      const int t16 = __builtin_amdgcn_readfirstlane((int)(grow0 >> 4));
      pool_e = pool_bits_s[t16];
      pool_s = pool_slot_s[t16];
    }

    // ---- accumulator init: bias ----------------------------------------------------------------
    f32x4 q0, q1, q2, q3;                        // 16 rows x 64 columns: four 16-column tiles
    f32x4 cs = {1.f, 1.f, 1.f, 1.f};             // F16: current scale of rows 4 (lane >> 4) + e of the accumulators
    float sc0 = 0.f, sc1 = 0.f;                  // F16: the same scales in the gather layout: rows g8 and 8 + g8
    (void)cs; (void)sc0; (void)sc1;              //      (0 = the row has had no non-zero block yet: its operands are zeros)
    if constexpr (F16) {
      q0 = q1 = q2 = q3 = f32x4{0.f, 0.f, 0.f, 0.f};         // (the bias joins in the epilogue, behind the scales)
    } else {
      const int c_ = lane & 15;
      const float b0_ = biasL[c_], b1_ = biasL[16 + c_], b2_ = biasL[32 + c_], b3_ = biasL[48 + c_];
      q0 = f32x4{b0_, b0_, b0_, b0_};
      q1 = f32x4{b1_, b1_, b1_, b1_};
      q2 = f32x4{b2_, b2_, b2_, b2_};
      q3 = f32x4{b3_, b3_, b3_, b3_};
    }
    // K blocks: b < KB-1 = relation slot b (gathered x rows), b == KB-1 = the row itself,
    // b == KB (ST > 0) = table pseudo block (gathered ytab rows, added in the C/D layout)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b < KB - 1 && !((live >> b) & 1)) continue;      // empty relation slot (wave-uniform)
      if (b >= KB && !(live & 0x300)) continue;            // no table source in the tile
      // ---- complete the gather of block b ------------------------------------------------------
      DESCO_ZERO_SUMS()
      if (b < KB - 1) {
        DESCO_FINISH(xb, LDX)
      } else if (b == KB - 1) {
        lo0 = DESCO_SELF_REG(0, 0); hi0 = DESCO_SELF_REG(0, 1); lo1 = DESCO_SELF_REG(1, 0); hi1 = DESCO_SELF_REG(1, 1);
      } else {
        // canonical->count relations have at most one source per row: one step covers both table
        // slots; anything beyond that (general inputs) takes the cooperative path
        int d0, d1, m0, m1;
        DESCO_CONSUME_TAB(0) DESCO_CONSUME_TAB(1)
        if (__any((c0 < n0) | (c1 < n1))) {
          DESCO_COOP(0, yb, LDY) DESCO_COOP(1, yb, LDY)
        }
        if (ST > 1 && __any((d0 < m0) | (d1 < m1))) {
          c0 = d0; c1 = d1;
          n0 = m0; n1 = m1;
          DESCO_COOP(0, yb + 64, LDY) DESCO_COOP(1, yb + 64, LDY)
        }
      }
      // (round 6: issuing them one block earlier, behind the gather of the last live relation block, measured 1-2 % SLOWER on
      //  all three shapes -- profiles/r6_j_ab_dma_early_*.log: the table step then queues behind them)
      if (b == KB - 1 && has_next && S > 0) {
        // CSR slice of the next tile, global -> LDS directly (no staging registers): lane i's dword lands
        // at rpn[i] / ecn[i].  Issued behind the last relation-slot block: vector memory returns in order,
        // and the on-demand steps of heavy rows should not queue behind these (slow) LDS-direct loads
        const int nptr = nrn * S + 1;
        const int32_t* src = g.vrowptr + (g.row0 + w0n) * S;
        DESCO_DMA4(src + (lane < nptr ? lane : nptr - 1), rpn);
        if (lane + 64 < nslot) DESCO_DMA4(src + (lane + 64 < nptr ? lane + 64 : nptr - 1), rpn + 64);
        const int32_t* ids = (g.vcol + ebn_cur) + (unsigned)lane;    // uniform base + 32-bit lane offset
        const int ne = (een_cur - ebn_cur) < WCAP ? (een_cur - ebn_cur) : WCAP;
#pragma unroll
        for (int k_ = 0; k_ < (WCAP + 63) / 64; ++k_)
          if (lane + 64 * k_ < ne) DESCO_DMA4(ids + 64 * k_, ecn + 64 * k_);
      }
      // ---- the two 32-column halves of block b; the first gather step of block b+1 goes out
      //      under this block's MFMAs (after the low halves have left their registers)
      {
        if (b < KB) {
          if constexpr (F16) {
            DESCO_BLOCK_SCALES()
            DESCO_PUT_F16(lo0, 0, sc0) DESCO_PUT_F16(lo1, 1, sc1)
          } else {
            DESCO_PUT_X6(lo0, 0) DESCO_PUT_X6(lo1, 1)
          }
        } else {
          DESCO_PUT_F32(lo0, 0) DESCO_PUT_F32(lo1, 1)
        }
      }
      DESCO_ISSUE_AFTER(b)
      {
        if (b >= KB) {
          if constexpr (F16) {
            DESCO_TAB_HALF_F16(q0, q1)
          } else {
            DESCO_TAB_HALF(q0, q1)
          }
        } else {
          if constexpr (F16) {
            DESCO_MFMA_HALF_F16(b, 0)
          } else {
            DESCO_MFMA_HALF_X6(b, 0)
          }
        }
        if (b < KB) {
          if constexpr (F16) {
            DESCO_PUT_F16(hi0, 0, sc0) DESCO_PUT_F16(hi1, 1, sc1)
          } else {
            DESCO_PUT_X6(hi0, 0) DESCO_PUT_X6(hi1, 1)
          }
        } else {
          DESCO_PUT_F32(hi0, 0) DESCO_PUT_F32(hi1, 1)
        }
        if (b >= KB) {
          if constexpr (F16) {
            DESCO_TAB_HALF_F16(q2, q3)
          } else {
            DESCO_TAB_HALF(q2, q3)
          }
        } else {
          if constexpr (F16) {
            DESCO_MFMA_HALF_F16(b, 1)
          } else {
            DESCO_MFMA_HALF_X6(b, 1)
          }
        }
      }
    }

    // ---- switch to the next tile: publish its source ids and launch its first gather step ------
    const int64_t grow_out = grow0;
    const int nr_out = nr;
    if (has_next) {
      // the next tile's row pointers and ids are in LDS (every gather of this tile has been consumed:
      // nothing else is outstanding, the wait orders the LDS reads behind the LDS-direct loads, which the
      // compiler does not track)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ec = ecn;
      cur ^= 1;
      rp = rpn;
      ebase = ebn_cur;
      sub_n1 = DESCO_NEXT_SUB();
      w0 = w0n;
      nr = nrn;
      grow0 = g.row0 + w0n;
      DESCO_TILE_LIVE()
      DESCO_ISSUE_AFTER(-1)
    }

    // ---- epilogue.  C/D map of a 16x16 tile: lane (c = lane&15, g = lane>>4) holds rows 4 g + e of
    //      column 16 t + c.  A 4x4 transpose over the lane quarters (tile t of quarter g <-> tile g of
    //      quarter t: v_permlane32_swap, then v_permlane16_swap) leaves lane = column with the 16 rows
    //      of the tile in registers, row 4 t + e in q_t[e]: every store is one full 256-byte row, and
    //      the pooling pass is a running sum in row order ------------------------------------------------
    if constexpr (F16) {
      // leave the scales (exact: powers of two) and add the bias: row 4 g + e of column 16 t + c
      const int c_ = lane & 15;
      const float b0_ = biasL[c_], b1_ = biasL[16 + c_], b2_ = biasL[32 + c_], b3_ = biasL[48 + c_];
      const f32x4 f_ = {pow2_inverse(cs[0]) * winv, pow2_inverse(cs[1]) * winv, pow2_inverse(cs[2]) * winv,
                        pow2_inverse(cs[3]) * winv};
      q0 = q0 * f_ + b0_;
      q1 = q1 * f_ + b1_;
      q2 = q2 * f_ + b2_;
      q3 = q3 * f_ + b3_;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      q0[e] = apply_act(q0[e], g.act, g.slope);
      q1[e] = apply_act(q1[e], g.act, g.slope);
      q2[e] = apply_act(q2[e], g.act, g.slope);
      q3[e] = apply_act(q3[e], g.act, g.slope);
    }
    if (g.row_absmax) {
      // per-row bound of the produced rows for a later f16x3 GEMM over them (the anchor operand): still in the C/D
      // layout -- lane (c, g) holds rows 4 g + e of columns 16 t + c --, so a row's maximum is 3 maxima over the tiles
      // and 4 DPP steps over the quarter's 16 lanes; accumulated over the launches that fill the operand's columns
      f32x4 m_ = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_abs(q0), __builtin_elementwise_abs(q1)),
                                           __builtin_elementwise_max(__builtin_elementwise_abs(q2), __builtin_elementwise_abs(q3)));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t v_ = __float_as_uint(m_[e]);
        uint32_t o_ = __builtin_amdgcn_update_dpp(0u, v_, 0xB1, 0xf, 0xf, true);
        v_ = v_ > o_ ? v_ : o_;
        o_ = __builtin_amdgcn_update_dpp(0u, v_, 0x4E, 0xf, 0xf, true);
        v_ = v_ > o_ ? v_ : o_;
        o_ = __builtin_amdgcn_update_dpp(0u, v_, 0x141, 0xf, 0xf, true);
        v_ = v_ > o_ ? v_ : o_;
        o_ = __builtin_amdgcn_update_dpp(0u, v_, 0x140, 0xf, 0xf, true);
        v_ = v_ > o_ ? v_ : o_;
        const int r_ = 4 * (lane >> 4) + e;
        if ((lane & 15) == 0 && r_ < nr_out) {
          float* p_ = g.row_absmax + (grow_out - g.row0) + r_;
          const float old_ = *p_;
          *p_ = fmaxf(old_, __uint_as_float(v_));
        }
      }
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
      if (g.out) {
        float* ob = g.out + grow_out * LDO + lane;               // LD64: row r at the immediate offset 256 r
        if (nru == 16) {
#pragma unroll
          for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(DESCO_ROW(r), ob + r * LDO);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (r < nru) __builtin_nontemporal_store(DESCO_ROW(r), ob + r * LDO);
        }
      }
      if (g.out2) {
        float* o2 = g.out2 + (grow_out - g.row0) * g.ldo2 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r < nru) o2[r * g.ldo2] = DESCO_ROW(r);
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
    if (!has_next) break;
  }
}


#undef DESCO_DMA4
#undef DESCO_NEXT_SUB
#undef DESCO_SUB_ROWS
#undef DESCO_CUR
#undef DESCO_CURS
#undef DESCO_ISSUE2
#undef DESCO_CONSUME2
#undef DESCO_ZERO_SUMS
#undef DESCO_ANY_STAGED
#undef DESCO_ISSUE_SELF
#undef DESCO_SELF_REG
#undef DESCO_TAB_CUR
#undef DESCO_ISSUE_TAB
#undef DESCO_CONSUME_TAB
#undef DESCO_COOP
#undef DESCO_FINISH
#undef DESCO_ISSUE_BLOCK
#undef DESCO_SLOT_ANY
#undef DESCO_SLOT_LIVE
#undef DESCO_TILE_LIVE
#undef DESCO_ISSUE_AFTER
#undef DESCO_PUT_F32
#undef DESCO_PUT_X6
#undef DESCO_M16
#undef DESCO_MFMA_HALF_X6
#undef DESCO_TAB_HALF
#undef DESCO_TAB_HALF_F16
#undef DESCO_MFMA_HALF_F16
#undef DESCO_F16
#undef DESCO_PUT_F16
#undef DESCO_BLOCK_SCALES
#undef DESCO_GROUP_MAX
#undef DESCO_ABSMAX8

template <int NW, int KB, int ST, bool LD64, bool POOL, bool F16, bool SELFDEG = false>
static void shmp16_launch_one(const ShmpArgs& g, unsigned grid, hipStream_t st) {
  constexpr int WST = KB * 64 + 16;
  constexpr size_t w_floats = (size_t)(F16 ? 2 : 3) * 64 * WST / 2;
  constexpr size_t shmem = sizeof(float) * (w_floats + (size_t)NW * (F16 ? WAVE_LDS_F16 : WAVE_LDS) + 64 + 4 +
                                           (SELFDEG ? (MAXS + 1) * 64 : 0));
  static_assert(shmem <= 160 * 1024, "SHMP layer (16-row tiles): LDS budget exceeded");
  static DeviceOnce attr_once;        // function attributes are per device
  if (!attr_once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shmp_layer16_kernel<NW, KB, ST, LD64, POOL, F16, SELFDEG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.mark();
  }
  hipLaunchKernelGGL((shmp_layer16_kernel<NW, KB, ST, LD64, POOL, F16, SELFDEG>), dim3(grid), dim3(NW * 64), shmem, st, g,
                     g.vrowptr, g.pool_bits, g.pool_slot);
}

template <int NW, int KB, bool F16>
static bool shmp16_launch_st(const ShmpArgs& g, unsigned grid, hipStream_t st) {
  const bool ld64 = g.ldx == 64 && (g.st == 0 || g.ldy == 64 * g.st) && (!g.out || g.ldo == 64);      // (xself has its own stride)
  if (g.pool_part) {
    if constexpr (KB == 3) {
      if (g.st != 2) return false;
      if (g.self_coef) {                 // the rows themselves recomputed from their slot degrees (fp16 form only)
        if constexpr (F16) {
          if (ld64)
            shmp16_launch_one<NW, 3, 2, true, true, true, true>(g, grid, st);
          else
            shmp16_launch_one<NW, 3, 2, false, true, true, true>(g, grid, st);
          return true;
        } else {
          return false;
        }
      }
      if (ld64)
        shmp16_launch_one<NW, 3, 2, true, true, F16>(g, grid, st);
      else
        shmp16_launch_one<NW, 3, 2, false, true, F16>(g, grid, st);
      return true;
    } else {
      return false;
    }
  }
  if (g.self_coef) return false;          // (built for the pooled count-row launch only)
#define DESCO_ONE(ST_)                                \
  if (ld64)                                           \
    shmp16_launch_one<NW, KB, ST_, true, false, F16>(g, grid, st);    \
  else                                                \
    shmp16_launch_one<NW, KB, ST_, false, false, F16>(g, grid, st);
  switch (g.st) {
    case 0: DESCO_ONE(0) break;
    case 1: DESCO_ONE(1) break;
    case 2: DESCO_ONE(2) break;
    default: return false;
  }
#undef DESCO_ONE
  return true;
}

template <int NW, bool F16>
static bool shmp16_launch_nw(const ShmpArgs& g, int cus, hipStream_t st) {
  const int64_t ntiles = (g.num_rows + NW * WR - 1) / (NW * WR);
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  switch (g.sm) {
    case 0: return shmp16_launch_st<NW, 1, F16>(g, grid, st);
    case 1: return shmp16_launch_st<NW, 2, F16>(g, grid, st);
    default: return shmp16_launch_st<NW, 3, F16>(g, grid, st);
  }
}

// x6 arguments validated by shmp_launch (shmp_layer.hip); g.wplanes set, g.sm <= 2
bool shmp16_launch(const ShmpArgs& g, int cus, void* stream) {
  if (!g.wplanes || g.sm < 0 || g.sm > 2 || g.S > MAXS) return false;
  // bf16x6 form: 16 waves per block (12 measured 2-6 % slower there: profiles/r2_h_ab_tile_rows.log).  fp16 form: TWELVE,
  // three per SIMD -- a quarter fewer tiles in flight per XCD means a quarter less traffic between two references to a
  // source row, and with half the matrix work per tile the fourth wave per SIMD is not needed to hide latency: Syn_1827
  // shapes 3.68 -> 3.39 ms per count-row launch, MSRC-21 + IMDB 2.93 -> 2.74, COX2 1.325 -> 1.30; 8, 10 and 14 waves are
  // all slower (profiles/r4_m_ab_shmp_waves.log)
  if (g.wscale) return shmp16_launch_nw<12, true>(g, cus, (hipStream_t)stream);     // fp16 three-product planes
  return shmp16_launch_nw<16, false>(g, cus, (hipStream_t)stream);
}

}  // namespace desco

