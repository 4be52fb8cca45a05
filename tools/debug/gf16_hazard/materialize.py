"""Rebuild the kernel sources the bisection harness compiles from what the repository keeps of them (no
superseded kernel copies in the tree):
    gossip_f16_var.hip               = desco_amd/csrc/gossip_f16.hip (the shipped kernel) + gossip_f16_var.patch
    old_0d06b19/gossip_wave_f16.hip  = `git show 0d06b19:desco_amd/csrc/gossip_f16.hip` edited by gossip_wave_f16.ed
                                       (a `diff -e` script: the block-form kernel of that commit cut, VAR_* switches added)
build_variants.sh calls this first; the rebuilt files are git-ignored."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))


def apply_ed(lines, script):
    """Apply a `diff -e` script (commands a / c / d, last hunk first, text ended by a lone '.')."""
    i, sc = 0, script
    while i < len(sc):
        m = re.fullmatch(r"(\d+)(?:,(\d+))?([acd])", sc[i].rstrip("\n"))
        assert m, sc[i]
        lo, hi, cmd = int(m.group(1)), int(m.group(2) or m.group(1)), m.group(3)
        i += 1
        text = []
        if cmd in "ac":
            while sc[i].rstrip("\n") != ".":
                text.append(sc[i])
                i += 1
            i += 1
        if cmd == "a":
            lines[lo:lo] = text
        elif cmd == "c":
            lines[lo - 1:hi] = text
        else:
            del lines[lo - 1:hi]
    return lines


def main():
    var = os.path.join(HERE, "gossip_f16_var.hip")
    subprocess.check_call(["cp", os.path.join(ROOT, "desco_amd", "csrc", "gossip_f16.hip"), var])
    subprocess.check_call(["patch", "-s", var, os.path.join(HERE, "gossip_f16_var.patch")])
    old = subprocess.check_output(["git", "-C", ROOT, "show", "0d06b19:desco_amd/csrc/gossip_f16.hip"]).decode()
    ed = open(os.path.join(HERE, "old_0d06b19", "gossip_wave_f16.ed")).readlines()
    out = apply_ed(old.splitlines(keepends=True), ed)
    open(os.path.join(HERE, "old_0d06b19", "gossip_wave_f16.hip"), "w").writelines(out)


if __name__ == "__main__":
    main()
