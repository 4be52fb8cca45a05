#!/usr/bin/env python3
"""Writes _build/shmp16_ldsgather.hip: shmp_layer16.hip with every staged gather (sources and the rows themselves) read from
an LDS image instead of global memory -- WRONG RESULTS by design (slot = row id mod RWN, nothing checks that the row is
there), timing only: the upper bound of what a window of x rows staged in LDS can buy the layer kernel (DESIGN.md 8,
round 5).  Every wave streams the 16 rows of its next tile into the image (LDS-direct loads), so the x rows are still
fetched once.  Switches: RWN (rows in the image, power of two), NWF (waves per block of the fp16 form)."""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "../../../desco_amd/csrc/shmp_layer16.hip")).read()


def rep(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old)
    src = src.replace(old, new)


rep("constexpr int WR = 16;        // rows per wave\n",
    "constexpr int WR = 16;        // rows per wave\n#ifndef RWN\n#define RWN 128\n#endif\n#ifndef NWF\n#define NWF 12\n#endif\n")
# sources of the relation-slot blocks and the rows themselves: from the LDS image
rep("    const float* p0_ = k0_ ? (base_) + (int64_t)i0_ * (ld_) : zrow;                   \\\n"
    "    const float* p1_ = k1_ ? (base_) + (int64_t)i1_ * (ld_) : zrow;                   \\\n",
    "    const float* p0_ = xw + ((k0_ ? i0_ & (RWN - 1) : RWN) << 6);                     \\\n"
    "    const float* p1_ = xw + ((k1_ ? i1_ & (RWN - 1) : RWN) << 6);                     \\\n")
rep("    const float* p_ = xb + (grow0 + (r_ < nr ? r_ : nr - 1)) * LDX;                          \\\n",
    "    const float* p_ = xw + (((int)(grow0 + (r_ < nr ? r_ : nr - 1)) & (RWN - 1)) << 6);       \\\n")
rep("  int* next_sub = reinterpret_cast<int*>(biasL + 64);      // the block's tile hand-out counter\n",
    "  int* next_sub = reinterpret_cast<int*>(biasL + 64);      // the block's tile hand-out counter\n"
    "  float* xwb = biasL + 64 + 4;                             // [RWN + 1][64] image of x rows (+ a zero row)\n"
    "  const float* xw = xwb + 4 * (lane & 7);\n"
    "  if (tid < 64) xwb[RWN * 64 + tid] = 0.f;\n")
# stream the next tile's rows into the image (4 LDS-direct loads of 1 KB)
rep("        const int32_t* ids = (g.vcol + ebn_cur) + (unsigned)lane;    // uniform base + 32-bit lane offset\n",
    "        {\n"
    "          const float* xs_ = g.x + (g.row0 + w0n) * 64 + 4 * lane;\n"
    "          float* xd_ = xwb + (((int)(g.row0 + w0n) & (RWN - 1)) << 6);\n"
    "          for (int q_ = 0; q_ < 4; ++q_)\n"
    "            if (4 * q_ < nrn)\n"
    "              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs_ + 256 * q_),\n"
    "                                               (__attribute__((address_space(3))) void*)(xd_ + 256 * q_), 16, 0, 0);\n"
    "        }\n"
    "        const int32_t* ids = (g.vcol + ebn_cur) + (unsigned)lane;    // uniform base + 32-bit lane offset\n")
rep("  constexpr size_t shmem = sizeof(float) * (w_floats + (size_t)NW * (F16 ? WAVE_LDS_F16 : WAVE_LDS) + 64 + 4);\n",
    "  constexpr size_t shmem = sizeof(float) * (w_floats + (size_t)NW * (F16 ? WAVE_LDS_F16 : WAVE_LDS) + 64 + 4 + (F16 ? (RWN + 1) * 64 : 0));\n")
rep("  if (g.wscale) return shmp16_launch_nw<12, true>(g, cus, (hipStream_t)stream);     // fp16 three-product planes\n",
    "  if (g.wscale) return shmp16_launch_nw<NWF, true>(g, cus, (hipStream_t)stream);     // fp16 three-product planes\n")
os.makedirs(os.path.join(here, "_build"), exist_ok=True)
open(os.path.join(here, "_build", "shmp16_ldsgather.hip"), "w").write(src)
