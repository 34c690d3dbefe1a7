#!/usr/bin/env python3
"""What the compiler made of the gather kernels' loads -- checked in the ISA, without a GPU.

The kernels of this library are bandwidth-bound: what they are worth is decided by how many row loads a wave keeps in
flight, and the source only ASKS for that (`UNROLL` loads, then the ANDs).  The scheduler is free to interleave, and its
register-pressure heuristics can quietly serialise a step's loads when it believes a higher occupancy is within reach --
round 5 lost 18-22 % on two row widths that way (profiles/r05_walk_occupancy_hint_ab.txt) and the count path's screen
launch ran with two or three loads in flight instead of eight, with no test failing.  This script counts, per kernel of
`make -C kwage_amd/csrc asm`'s gfx950 assembly, the runs of consecutive 16-byte loads between two `s_waitcnt vmcnt`, and
the scratch (spill) bytes, and compares them with what the source asks for.

    python tools/isa_check.py [--build] [--all]        (exit status 1 on a violation; tests/test_kernel_isa.py calls check())"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "kwage_amd", "lib", "asm", "engine-hip-amdgcn-amd-amdhsa-gfx950.s")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "kwage_amd", "csrc"), "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def kernels(path=ASM):
    """-> {(kernel, template ints): (Counter of load-run lengths, scratch bytes per lane)}"""
    s = open(path).read()
    scratch = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.amdhsa_kernel (\w+)\n(?:.*\n)*?\s*\.amdhsa_private_segment_fixed_size (\d+)", s)}
    out = {}
    for m in re.finditer(r"^(_ZN5kwage\w+):", s, re.M):
        name = m.group(1)
        body = s[m.end():s.find(".Lfunc_end", m.end())]
        runs, cur = [], 0
        for ln in body.splitlines():
            ln = ln.strip()
            if re.match(r"(buffer|global)_load_dwordx4", ln):
                cur += 1
            elif ln.startswith("s_waitcnt") and "vmcnt" in ln:
                if cur:
                    runs.append(cur)
                cur = 0
        d = re.match(r"_ZN5kwage(?:\d+_GLOBAL__N_1)?\d+(\w+?)(?:I(.*?)E)?Ev", name)
        key = (d.group(1), tuple(int(a) for a in re.findall(r"L[ib](\d+)E", d.group(2) or ""))) if d else (name, ())
        out[key] = (Counter(runs), scratch.get(name, 0))
    return out


def violations(ks):
    bad = []

    def need(key, ok, what):
        if key in ks and not ok(ks[key][0]):
            bad.append("%s<%s>: %s; runs of loads in flight: %s" % (key[0], ",".join(map(str, key[1])), what, dict(ks[key][0])))
    for (name, a), (runs, scratch) in ks.items():
        # (32 counter planes -- queries above 2^20 positions -- have spilled a few dozen dwords since round 3: known, bounded)
        known = (name == "count_walk_kernel" and ((a[0] == 32 and scratch <= 320) or (len(a) == 3 and a[2] == 1 and scratch <= 64)))      # (TRUNC: a few dwords in the hand-over path)
        if scratch and name.endswith("_kernel") and not known:
            bad.append("%s<%s>: %d bytes of scratch per lane (register spills)" % (name, ",".join(map(str, a)), scratch))
        if name == "and_walk_kernel":
            ch, u = a
            need((name, a), lambda r: sum(v for k, v in r.items() if k >= u - (1 if ch == 1 else 0)) >= ch, "every KiB-step should request its %d rows together" % u)
        elif name == "and_band_walk_kernel":
            ch, u = a
            need((name, a), lambda r: r.get(u, 0) >= ch, "every KiB-step should request its %d rows together" % u)
        elif name == "and_screen_kernel":
            vec, u = a
            need((name, a), lambda r: max(r, default=0) >= vec * u, "a tile's %d loads should be in flight together" % (vec * u))
        elif name == "and_refine_kernel":
            need((name, a), lambda r: r.get(a[0], 0) >= 64 // a[0], "%d rows in flight per 128-byte group" % a[0])
        elif name == "count_walk_kernel":
            planes, nh = a[:2]          # (the third argument: TRUNC)
            trunc = len(a) == 3 and a[2] == 1
            need((name, a), lambda r: max(r, default=0) >= (nh if (trunc and planes >= 20) else min(4 * nh, 8)), "at least eight rows (four k-mers' with one hash) in flight")
        elif name == "count_screen_kernel" and a[0] <= 10:
            planes, nh = a
            need((name, a), lambda r: max(r, default=0) >= min(8, 2 * nh + (2 if nh == 1 else 0)) and (nh != 1 or max(r, default=0) >= 8), "eight rows in flight (one hash), two k-mers' rows at least otherwise")
        elif name == "count_refine_kernel" and a[1] == 7:
            need((name, a), lambda r: max(r, default=0) >= 8, "at least eight rows in flight")
    return bad


def check(rebuild=True):
    if rebuild or not os.path.exists(ASM):
        build()
    ks = kernels()
    return ks, violations(ks)


if __name__ == "__main__":
    ks, bad = check("--build" in sys.argv or not os.path.exists(ASM))
    if "--all" in sys.argv:
        for (name, a), (runs, scratch) in sorted(ks.items(), key=str):
            print("%s<%s>  loads in flight %s%s" % (name, ",".join(map(str, a)), dict(runs), ("  SCRATCH %d" % scratch) if scratch else ""))
    print("%d kernels, %d violations" % (len(ks), len(bad)))
    for b in bad:
        print("  " + b)
    sys.exit(1 if bad else 0)
