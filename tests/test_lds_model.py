"""The LDS layout of the Winograd-x conv against the hardware's ds_read_b128 lane groups (tools/lds_model.py; MI355X_MICROARCH.md, LDS):
the layout the kernel source uses must be the conflict-free one of the model, and the model must still call the rounds 3-4 layout what the
SQ counters called it (every A-fragment read a two-way conflict)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lds_model as m          # noqa: E402


def test_round5_layout_reads_conflict_free():
    reads, stores = m.score((0, 1, 3, 2), (0, 0, 2, 2))
    assert reads == 0
    old_reads, old_stores = m.score((0, 1, 2, 3), (0, 1, 2, 3))
    assert old_reads == 160            # 40 reads x 4 extra cycles: two-way on every one
    assert stores <= old_stores        # the record stores are no worse than before
    assert all(m.cycles_read_b128(m.b_read(lo)) == 4 for lo in (0, 1))


def test_kernel_source_uses_the_modelled_layout():
    src = open(os.path.join(ROOT, "3d-condtional-stable-diffusion_amd", "csrc", "dm3d_conv_h3w.hip")).read()
    assert re.search(r"xq = \(row >> 2\) \^ \(row >> 3\)", src), "fragment row group -> x-pair g ^ (g >> 1)"
    assert re.search(r"\(q \^ \(\(ay \+ dy\) & 2\)\) << 4", src), "fragment slot swizzle y & 2"
    assert re.search(r"\(piece \^ \(\(srow % HH\) & 2\)\) << 3", src), "store slot swizzle y & 2"
    assert re.search(r"x = 2 \* \(\(row_t >> 2\) \^ \(row_t >> 3\)\)", src), "skip tail rows follow the same x-pair order"
    parts = open(os.path.join(ROOT, "3d-condtional-stable-diffusion_amd", "csrc", "dm3d_conv_h3v2_parts.h")).read()
    assert re.search(r"xp = \(lane >> 4\) \^ \(lane >> 5\)", parts), "epilogue_cq reads the accumulator row groups in that order"


def test_lane_groups_partition_the_wave():
    lanes = sorted(l for g in m.G128 for l in g)
    assert lanes == list(range(64)) and all(len(g) == 16 for g in m.G128)
