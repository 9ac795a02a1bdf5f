"""Round-5 golden fixture, produced by running the REFERENCE's own build_dfm_cost on PyTorch-CPU.

Build container only (needs /root/reference):   python tests/golden/make_golden_r05.py
Same rules as make_golden.py: the reference function is lifted by AST and executed unmodified; only
the inputs and the outputs it produced are stored.

  plane_sweep_zero_depth.npz   a sweep that REACHES NON-FINITE SAMPLING COORDINATES: cur2prev is a pure
      translation by -depths[1] along the optical axis and the intrinsics have no fourth column, so every
      lattice point of plane 1 lands at z = 0 in the previous camera and the reference's unguarded
      projection (core/bbox/structures/utils.py:209) divides by zero: x / 0 = +-Inf.  (NO coordinate of this
      fixture is NaN: with f = 720 the principal-axis pixel does not un-project to exactly 0 in fp32 -- the
      generator prints `of which NaN: 0`.  make_golden_r06.py's plane_sweep_nan_coords.npz reaches 0 / 0.)
      F.grid_sample on PyTorch-CPU answers such coordinates with NaN (its bilinear weights
      are Inf - Inf); the oracle and the HIP kernels answer 0 -- the value `padding_mode='zeros'` gives
      every other coordinate outside the map.  The fixture stores the reference's output (NaNs included)
      and its grids; tests/test_oracle_golden.py and tests/test_plane_sweep_gpu.py pin both facts: equal
      bits wherever the reference is finite, exactly 0 where it is NaN, and the NaNs are exactly the
      points whose reference grid is non-finite.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    import make_golden as g1
    g = g1.load_reference()
    C, H, W, D, fsf, csf = 4, 24, 78, 4, 16, 1
    gen = torch.Generator().manual_seed(500)
    cur = torch.randn(1, C, H, W, generator=gen)
    prev = torch.randn(1, C, H, W, generator=gen)
    dmin, dmax = 2.0, 6.0
    depths = torch.tensor([dmin + (k + 0.5) * ((dmax - dmin) / D) for k in range(D)], dtype=torch.float32)
    # intrinsics without the fourth column (projected depth == z), principal point ON a lattice pixel
    # (lattice x = 16 * 38 = 608, y = 16 * 11 = 176): 0 / 0 there, +-Inf elsewhere on plane 1
    P = torch.tensor([[720.0, 0, 608.0, 0], [0, 720.0, 176.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
    Tm = torch.eye(4)
    Tm[2, 3] = -float(depths[1])
    captured = []
    orig = F.grid_sample

    def capture(inp, grid, **kw):
        captured.append(grid.clone())
        return orig(inp, grid, **kw)
    F.grid_sample = capture
    try:
        out = g['build_dfm_cost'](cur, prev, depths, fsf, csf, P[None], Tm[None], (375, 1242), False, (0, 0), 1.0)
    finally:
        F.grid_sample = orig
    pg = captured[1].numpy().reshape(-1, 2)
    print('output', tuple(out.shape), 'NaN outputs:', int(torch.isnan(out).sum()),
          'non-finite prev grid points:', int((~np.isfinite(pg).all(1)).sum()),
          'of which NaN:', int(np.isnan(pg).any(1).sum()))
    np.savez_compressed(
        os.path.join(HERE, 'plane_sweep_zero_depth.npz'),
        cur=cur.numpy(), prev=prev.numpy(), depths=depths.numpy(), P=P.numpy(), Pinv=torch.inverse(P).numpy(),
        T=Tm.numpy(), fsf=np.float64(fsf), csf=np.float64(csf), flip=np.bool_(False),
        crop=np.asarray((0, 0), np.float64), scale=np.float64(1.0), img_shape=np.asarray((375, 1242)),
        ref_out=out.numpy(), ref_cur_grid=captured[0].numpy().reshape(-1, 2), ref_prev_grid=pg)


if __name__ == '__main__':
    main()
