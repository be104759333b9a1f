"""Storage of the polish's inverse factor (kernels/lh_layout.h, used by kernels/lh_inverse.inc): the index functions are plain C, so their
invariants are checked on the CPU -- rows do not overlap, a row of group g holds 8 (g + 1) entries at an odd stride, lhp_size covers the
last row, and (row 0, column 1) -- the stored zero the deletion reads for masked lanes -- lies inside row 0's padding."""
import os
import subprocess
import tempfile

SRC = r"""
#define __host__
#define __device__
#include "lh_layout.h"
#include <cstdio>
#include <initializer_list>
int main() {
    for (int rows : {8, 112, 160, 192}) {
        int prev_end = 0;
        for (int b = 0; b < rows; ++b) {
            const int g = b >> 3, off = lhp_row(b), len = lhp_len(g);
            if (len != 8 * (g + 1) || len <= b) return 1;                 // the diagonal entry fits
            if (off < prev_end) return 2;                                // no overlap with the previous row
            if (b & 7) { if (off - lhp_row(b - 1) != len + 1) return 3; } // odd stride inside a group
            else if (off != lhp_grp(g)) return 4;
            prev_end = off + len;
        }
        if (prev_end > lhp_size(rows)) return 5;
        if (lhp_size(rows) != lhp_grp(rows / 8)) return 6;
    }
    if (!(lhp_row(0) + 1 < lhp_row(0) + lhp_len(0))) return 7;
    std::printf("ok %d %d\n", lhp_size(112), lhp_size(192));
    return 0;
}
"""


def test_layout_of_the_inverse_factor():
    here = os.path.dirname(os.path.abspath(__file__))
    inc = os.path.join(here, "..", "swarm_simulator_amd", "csrc", "kernels")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(SRC)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", inc, "-o", exe, src])
        out = subprocess.check_output([exe]).decode()
    assert out.split() == ["ok", "6832", "19392"]
