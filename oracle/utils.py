"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's data-preparation helper (never imported by the product).

spatial_coarsen: /root/reference/src/utils.py:13-60.  Pinned by tests/golden/utils_coarsen.npz (captured from the
reference by oracle/capture.py, worker `utils`), bitwise.
"""
import numpy as np


def spatial_coarsen(X, Y, u_seq, v_seq, p_seq, agg_x=4, agg_y=4):
    """Block means over agg_x x agg_y cells (src/utils.py:37-58), vectorised: the cells of a block are laid out as the
    trailing contiguous axis in the reference's flattening order ([agg_x][agg_y] row-major, :51-53), so numpy.mean
    reduces them with the same pairwise add order as the reference's per-block call.

    Reference behaviour kept: the new meshgrid uses numpy's default 'xy' indexing on linspace(0, 2, .) (:45-47), so it
    has shape [ny/agg_y, nx/agg_x]; the column loop runs over ny // agg_x blocks (:49): fewer than ny // agg_y leaves
    the remaining cells 0, more overruns the output (IndexError at :56)."""
    nx, ny = X.shape[0], X.shape[1]
    T = u_seq.shape[0]
    assert nx % agg_x == 0
    assert ny % agg_y == 0
    cnx, cny = nx // agg_x, ny // agg_y
    jfill = ny // agg_x
    if jfill > cny:
        raise IndexError("index %d is out of bounds for axis 2 with size %d" % (cny, cny))
    new_X, new_Y = np.meshgrid(np.linspace(0, 2, cnx), np.linspace(0, 2, cny))

    def coarsen(f):
        blocks = np.ascontiguousarray(np.asarray(f).reshape(T, cnx, agg_x, cny, agg_y).transpose(0, 1, 3, 2, 4)).reshape(T, cnx, cny, agg_x * agg_y)
        out = np.zeros((T, cnx, cny))
        out[:, :, :jfill] = np.mean(blocks[:, :, :jfill], axis=-1)
        return out

    return new_X, new_Y, coarsen(u_seq), coarsen(v_seq), coarsen(p_seq)
