#!/usr/bin/env python3
"""Writes _build/shmp16_parts.hip: the shipped shmp_layer16.hip with timing-only switches (wrong results by design) that
remove one part of a tile's work each:  ABL_NOTAB  no table pseudo block (canonical -> count sources)   ABL_NOSLOTS  no
relation-slot blocks (the row itself and the table only)   ABL_NOMFMA  no MFMA issued   ABL_NOSTORE  no output rows stored
ABL_NOSCALE  row scales fixed at 1 (no maxima, no votes, no accumulator rescaling)"""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "../../../desco_amd/csrc/shmp_layer16.hip")).read()


def rep(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old)
    src = src.replace(old, new)


rep("    if (ST > 1) live |= DESCO_SLOT_ANY(g.sm + 1) ? 0x200 : 0;    \\\n  }\n",
    "    if (ST > 1) live |= DESCO_SLOT_ANY(g.sm + 1) ? 0x200 : 0;    \\\n    DESCO_ABL_LIVE                                               \\\n  }\n")
rep("#define DESCO_TILE_LIVE()                                        \\\n",
    "#if defined(ABL_NOTAB)\n#define DESCO_ABL_LIVE live &= 0xff;\n#elif defined(ABL_NOSLOTS)\n#define DESCO_ABL_LIVE live &= 0x300;\n"
    "#else\n#define DESCO_ABL_LIVE\n#endif\n#define DESCO_TILE_LIVE()                                        \\\n")
rep("#define DESCO_F16(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0);",
    "#if defined(ABL_NOMFMA)\n#define DESCO_F16(a_, b_, c_) asm volatile(\"\" : \"+v\"(c_) : \"v\"(a_), \"v\"(b_));\n#else\n"
    "#define DESCO_F16(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0);\n#endif")
rep("    if (!POOL || g.out) {\n      float* ob = g.out + grow_out * LDO + lane;",
    "#if defined(ABL_NOSTORE)\n    if (g.num_rows < 0) {\n#else\n    if (!POOL || g.out) {\n#endif\n      float* ob = g.out + grow_out * LDO + lane;")
rep("    const bool o0_ = m0_ * sc0 > 60000.f || (sc0 == 0.f && m0_ > 0.f);                                          \\\n"
    "    const bool o1_ = m1_ * sc1 > 60000.f || (sc1 == 0.f && m1_ > 0.f);                                          \\\n",
    "    DESCO_ABL_SCALE                                                                                             \\\n"
    "    const bool o0_ = m0_ * sc0 > 60000.f || (sc0 == 0.f && m0_ > 0.f);                                          \\\n"
    "    const bool o1_ = m1_ * sc1 > 60000.f || (sc1 == 0.f && m1_ > 0.f);                                          \\\n")
rep("#define DESCO_BLOCK_SCALES()                                                                                    \\\n",
    "#if defined(ABL_NOSCALE)\n#define DESCO_ABL_SCALE sc0 = sc1 = 1.f; m0_ = m1_ = 0.f;\n#else\n#define DESCO_ABL_SCALE\n#endif\n"
    "#define DESCO_BLOCK_SCALES()                                                                                    \\\n")
os.makedirs(os.path.join(here, "_build"), exist_ok=True)
open(os.path.join(here, "_build", "shmp16_parts.hip"), "w").write(src)
