"""GPU: the hand-off granules of the persistent decode kernel (csrc/handoff.h) are never seen torn.

decode_persist.hip accepts a value as soon as the 8-byte {tag, value} pair that carries it shows the producing stage's tag;
that is only sound if an aligned 8-byte pair written by ONE write-through store (or two pairs by one 16-byte store) is never
observed half-written by an L1-bypassing load on another XCD.  whisper-burn_amd/tools/handoff_stress.cpp (built by
csrc/Makefile next to the library) hammers exactly those store / load instructions across XCDs: 128 producer blocks
rewriting their granules as fast as they can, 128 consumer blocks on other XCDs checking value == mix(tag) on every read,
for 2 x 10^10 granule reads per mode (8-byte stores; 16-byte stores read by 8-byte loads; 16-byte stores read by 16-byte loads)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "whisper-burn_amd", "lib", "handoff_stress")


@pytest.mark.gpu
def test_granules_are_never_torn_across_xcds():
    assert os.path.exists(BIN), "build the stress tool first: make -C whisper-burn_amd/csrc stress"
    # 20 000 million granule reads per mode (~0.1 - 0.3 s each: the readers pull ~5 TB/s of 8-byte granules)
    p = subprocess.run([BIN, "20000"], capture_output=True, text=True, timeout=300)
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    print(p.stdout)
    assert p.returncode == 0 and "HANDOFF_STRESS_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    assert len(rows) == 6                                 # three store / load widths, readers flat out and napping
    for r in rows:
        assert r["torn"] == 0 and r["backwards"] == 0, r
        # the readers did see the writers' values change under them (r04_d: 8 000 - 25 000 changes per flat-out pass -- the
        # readers' 4 TB/s of granule loads leave the write-through stores little of the fabric)
        assert r["tag_changes_seen"] >= (1000 if r["readers"] == "flat out" else 1), r
        if r["readers"] == "flat out":
            assert r["granules_read"] >= 1_000_000_000, r


def test_stress_tool_is_built_with_the_library():
    """(CPU) `make` in csrc/ -- what __graft_entry__.build() runs -- produces the stress binary next to the library."""
    assert os.path.exists(BIN) and os.access(BIN, os.X_OK)
