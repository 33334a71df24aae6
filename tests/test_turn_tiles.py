"""The wave-private LDS transposes of the training chains (csrc/fchain.h: store_feat_lines / turn_in) and of the streamed edge
kernel's whole-line epilogue (csrc/hmlp.hip: HM_LINES) write a 32 x 32 tile in one layout and read it back in the other with no
barrier or fence in between.  Hardware: one wave's LDS instructions execute in order.  Compiler: it may reorder a read above a write
only if it can prove that the two addresses never coincide for a thread -- so the property that keeps the order is that EVERY
read instruction of a turn has a lane that reads a 16-byte piece that same lane wrote, and every write instruction of the next turn a
lane that overwrites a piece it has just read.  This test restates the index maps of the kernels and checks exactly that
(round-5 advisor finding; fences were measured and cost the training step 2.8 %)."""
TURN_LD = 36   # floats per tile row (fchain.h TURN_LD, hmlp.hip HM_TURN_LD)


def acc_piece(lane, g):
    """accumulator layout: lane (n = lane & 31, hi = lane >> 5) holds, of row n, the four floats at column 8 g + 4 hi"""
    return (lane & 31) * TURN_LD + 4 * (lane >> 5) + 8 * g


def row_piece(lane, j):
    """row-major layout: lane (rr = lane >> 3, cq = lane & 7) holds, of row rr + 8 j, the four floats at column 4 cq"""
    return ((lane >> 3) + 8 * j) * TURN_LD + 4 * (lane & 7)


def overlaps(a, b):   # two 4-float pieces
    return abs(a - b) < 4


def check(write_piece, read_piece):
    for r in range(4):    # every read instruction has a lane that reads what it wrote itself ...
        assert any(overlaps(read_piece(l, r), write_piece(l, w)) for l in range(64) for w in range(4)), ("read", r)
    for w in range(4):    # ... and every write of the next turn a lane that overwrites what it has just read
        assert any(overlaps(write_piece(l, w), read_piece(l, r)) for l in range(64) for r in range(4)), ("write", w)
    # the two layouts cover the same 32 x 32 floats exactly once each (it is a transposition of pieces, nothing is lost)
    for piece in (write_piece, read_piece):
        cells = sorted(piece(l, q) + t for l in range(64) for q in range(4) for t in range(4))
        assert cells == sorted(r * TURN_LD + c for r in range(32) for c in range(32))


def test_store_feat_lines_turn_keeps_program_order():
    check(acc_piece, row_piece)     # accumulator layout in, whole lines out (store_feat_lines; HM_LINES e + e' stores)


def test_turn_in_keeps_program_order():
    check(row_piece, acc_piece)     # whole lines in, accumulator layout out (turn_in / load_feat_lines; HM_LINES residual rows)


def test_the_kernels_use_these_index_maps():
    """The maps above are the kernels': guards against the sources drifting away from what this test checks."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = open(os.path.join(root, "gnn_manip_amd", "csrc", "fchain.h")).read()
    assert "constexpr int TURN_LD = 36;" in f
    assert "float* wr = turn + n * TURN_LD + 4 * hi;" in f and "const float* rd = turn + rr * TURN_LD + 4 * cq;" in f      # store_feat_lines
    assert "float* wr = turn + rr * TURN_LD + 4 * cq;" in f and "const float* rd = turn + n * TURN_LD + 4 * hi;" in f      # turn_in
    assert "wr + 8 * g" in f and "rd + 8 * j * TURN_LD" in f and "wr + 8 * j * TURN_LD" in f and "rd + 8 * g" in f
    h = open(os.path.join(root, "gnn_manip_amd", "csrc", "hmlp.hip")).read()
    assert "constexpr int HM_TURN_LD = 36;" in h
    assert "float* t_acc = turn + n * HM_TURN_LD + 4 * hi;" in h
    assert "float* t_row = turn + (lane >> 3) * HM_TURN_LD + 4 * (lane & 7);" in h
    assert "t_acc + 8 * g" in h and "t_row + 8 * j * HM_TURN_LD" in h
