"""TEST INFRASTRUCTURE (never imported by the product path).

NumPy restatement of the counter-based generator behind the HIP path's random
inputs: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers:
as easy as 1, 2, 3", SC'11; the Random123 library).  The reference draws its
noise with TensorFlow's stateful ops (``q_z_given_x.sample``, va:2363;
``tf.nn.dropout``, mu:45-50), whose streams are not reproducible across runs;
the build makes every draw a pure function of (seed, stream, row, column) so
that a step can be replayed.  This module pins that function:

* ``philox4x32_10`` is checked against the known-answer vectors published with
  Random123 (tests/test_oracle_kat.py);
* ``dropout_mask`` / ``standard_normal`` restate how the kernels map counters
  to draws (scvae_amd/csrc/elementwise.hip: dropout_apply_kernel,
  philox_normal_kernel) and are compared with the device output bit for bit
  (masks) / to float rounding (Box-Muller) in tests/test_gpu_ops.py.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """``counter``: [..., 4] uint32, ``key``: [..., 2] uint32 (broadcastable)
    -> [..., 4] uint32 after ten rounds."""
    counter = np.asarray(counter, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0 = key[..., 0].astype(np.uint64)
    k1 = key[..., 1].astype(np.uint64)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & MASK32,
             (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & MASK32]
        k0 = (k0 + np.uint64(W0)) & MASK32
        k1 = (k1 + np.uint64(W1)) & MASK32
    return np.stack(c, axis=-1).astype(np.uint32)


def _draws(rows, cols, row_offset, seed, word3):
    """Raw 32-bit draws [rows, cols]: counter = (row_lo, row_hi, column // 4,
    word3), key = (seed_lo, seed_hi), element column % 4 of the output."""
    groups = (cols + 3) // 4
    r = (np.arange(rows, dtype=np.uint64) + np.uint64(row_offset))[:, None]
    g = np.arange(groups, dtype=np.uint64)[None, :]
    counter = np.stack(np.broadcast_arrays(
        r & MASK32, r >> np.uint64(32), g, np.uint64(word3)), axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF],
                   dtype=np.uint64)
    out = philox4x32_10(counter.astype(np.uint32), key.astype(np.uint32))
    return out.reshape(rows, groups * 4)[:, :cols]


def uniform(bits):
    """((x >> 8) + 0.5) * 2^-24 in float32, as the kernels compute it."""
    return ((bits >> np.uint32(8)).astype(np.float32)
            + np.float32(0.5)) * np.float32(2.0 ** -24)


def dropout_mask(rows, cols, keep, seed, site):
    """``mask / keep`` [rows, cols] float32 of scvae_dropout_apply: an
    element is kept iff its uniform draw is below ``keep``."""
    u = uniform(_draws(rows, cols, 0, seed, 0x80000000 | site))
    inv = np.float32(1.0) / np.float32(keep)
    return np.where(u < np.float32(keep), inv, np.float32(0.0)).astype(
        np.float32)


def standard_normal(rows, cols, row_offset, seed, stream_id):
    """scvae_philox_normal: Box-Muller on the uniform pairs (0, 1), (2, 3) of
    every counter."""
    groups = (cols + 3) // 4
    # 64-bit stream id: low word -> counter word 3, high word folded into key 1
    stream_id = int(stream_id)
    seed = int(seed) ^ ((stream_id >> 32) << 32)
    u = uniform(_draws(rows, groups * 4, row_offset, seed,
                       stream_id & 0xFFFFFFFF))
    u = u.reshape(rows, groups, 4).astype(np.float64)
    r0 = np.sqrt(-2.0 * np.log(u[..., 0]))
    r1 = np.sqrt(-2.0 * np.log(u[..., 2]))
    t0 = 2.0 * np.pi * u[..., 1]
    t1 = 2.0 * np.pi * u[..., 3]
    n = np.stack([r0 * np.cos(t0), r0 * np.sin(t0),
                  r1 * np.cos(t1), r1 * np.sin(t1)], axis=-1)
    return n.reshape(rows, groups * 4)[:, :cols]
