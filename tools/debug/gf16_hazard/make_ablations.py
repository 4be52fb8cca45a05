#!/usr/bin/env python3
"""Writes _build/gossip_f16_abl.hip: the shipped kernel (desco_amd/csrc/gossip_f16.hip) with timing-ablation switches
patched in (wrong results by design; developer tool -- the shipped source carries no ablation code):
  ABL_NOMFMA   no MFMA is issued          ABL_NOREQ   no weight-fragment LDS reads (the ring keeps its first contents)
  ABL_NOP1     no neighbour steps          ABL_NOEPI   no epilogue arithmetic between the blocks (fragments rebuilt from
  ABL_NOHEAD   blocks 5..8 and the head skipped          the raw accumulators)"""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "../../../desco_amd/csrc/gossip_f16.hip")).read()


def rep(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old)
    src = src.replace(old, new)


rep('#define GF16_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0);',
    '#if defined(ABL_NOMFMA)\n#define GF16_MFMA(a_, b_, c_) asm volatile("" : "+v"(c_) : "v"(a_), "v"(b_));\n#else\n'
    '#define GF16_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0);\n#endif')
rep('#define GF16_RD_(dst_, base_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(base_), "n"(off_));',
    '#if defined(ABL_NOREQ)\n#define GF16_RD_(dst_, base_, off_) asm volatile("" : "+v"(dst_) : "v"(base_));\n#else\n'
    '#define GF16_RD_(dst_, base_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(base_), "n"(off_));\n#endif')
rep('    maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);',
    '    maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);\n#if defined(ABL_NOP1)\n    maxdeg = 0;\n#endif')
for n in (1, 2, 3):
    line = f"        GW_EPI{n}(acc0, 0) GW_EPI{n}(acc1, 1) GW_EPI{n}(acc2, 2) GW_EPI{n}(acc3, 3)\n"
    rep(line, "#if !defined(ABL_NOEPI)\n" + line + "#endif\n")
rep("      GW_HEAD(0) GW_HEAD(1) GW_HEAD(2) GW_HEAD(3)\n",
    "#if !defined(ABL_NOHEAD)\n      GW_HEAD(0) GW_HEAD(1) GW_HEAD(2) GW_HEAD(3)\n#endif\n")
# scheduling experiments (right results): EXP_PRIO=1 raises the wave's priority for the MFMA blocks, EXP_PRIO=2 for the
# VALU phases instead; EXP_STAGGER delays waves 4-7 by about half a query before their first unit
rep("#define GF16_BLOCK(b_, X_)                                                                              \\\n",
    "#if defined(EXP_PRIO) && EXP_PRIO == 1\n#define GF16_PRIO_ON __builtin_amdgcn_s_setprio(1);\n#define GF16_PRIO_OFF __builtin_amdgcn_s_setprio(0);\n"
    "#elif defined(EXP_PRIO) && EXP_PRIO == 2\n#define GF16_PRIO_ON __builtin_amdgcn_s_setprio(0);\n#define GF16_PRIO_OFF __builtin_amdgcn_s_setprio(1);\n"
    "#else\n#define GF16_PRIO_ON\n#define GF16_PRIO_OFF\n#endif\n"
    "#define GF16_BLOCK(b_, X_)                                                                              \\\n  GF16_PRIO_ON \\\n")
rep("  GF16_PAIR(4 * (b_) + 2, acc2, acc3, X_.h0, X_.l0) GF16_PAIR(4 * (b_) + 3, acc2, acc3, X_.h1, X_.l1)\n",
    "  GF16_PAIR(4 * (b_) + 2, acc2, acc3, X_.h0, X_.l0) GF16_PAIR(4 * (b_) + 3, acc2, acc3, X_.h1, X_.l1) GF16_PRIO_OFF\n")
rep("  while (unit < nunits) {\n",
    "#if defined(EXP_STAGGER)\n  if (wave >= 4) __builtin_amdgcn_s_sleep(100);\n#endif\n  while (unit < nunits) {\n")
open(os.path.join(here, "_build", "gossip_f16_abl.hip"), "w").write(src)
