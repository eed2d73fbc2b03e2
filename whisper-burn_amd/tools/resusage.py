#!/usr/bin/env python3
"""Print VGPR / scratch / occupancy / LDS per kernel from `hipcc -Rpass-analysis=kernel-resource-usage` output."""
import re
import sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split(" [")[0]
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"
    if re.search(pat, name):
        short = re.sub(r"_ZN2wb\d+_GLOBAL__N_1", "", name)[:64]
        print("%-64s VGPR %s AGPR %s SGPR %s scratch %s occ %s LDS %s" % (
            short, g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"),
            g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
