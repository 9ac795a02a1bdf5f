"""Round-6 golden fixture, produced by running the REFERENCE's own build_dfm_cost on PyTorch-CPU.

Build container only (needs /root/reference):   python tests/golden/make_golden_r06.py
Same rules as make_golden.py: the reference function is lifted by AST and executed unmodified; only
the inputs and the outputs it produced are stored.

  plane_sweep_nan_coords.npz   a sweep whose sampling coordinates reach NaN (0 / 0), not only +-Inf.
      plane_sweep_zero_depth.npz (round 5) puts plane 1 at z = 0 in the previous camera, but with f = 720 the
      un-projection of the principal-axis pixel is not EXACTLY zero in fp32 (608 d / 720 - d 608 / 720 leaves a
      rounding residue), so every bad coordinate there is x / 0 = +-Inf and none is NaN (its generator prints
      `of which NaN: 0`).  Here every factor is a dyadic rational -- f = 512, principal point (608, 176) ON a
      lattice pixel, depths[1] = 3.5 -- so the lattice column u = 608 un-projects to x_cam = 0 exactly and the
      lattice row v = 176 to y_cam = 0 exactly; on plane 1 (z = 0 in the previous camera) their projections
      are 0 / 0 = NaN: a whole column with a NaN x, a whole row with a NaN y, both at the crossing, +-Inf on
      the rest of the plane.  F.grid_sample on PyTorch-CPU answers all of them with NaN; the oracle and the HIP
      kernels with +0 (tests/util.assert_matches_reference).  The generator ASSERTS that NaN coordinates occur.
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
    gen = torch.Generator().manual_seed(600)
    cur = torch.randn(1, C, H, W, generator=gen)
    prev = torch.randn(1, C, H, W, generator=gen)
    dmin, dmax = 2.0, 6.0
    depths = torch.tensor([dmin + (k + 0.5) * ((dmax - dmin) / D) for k in range(D)], dtype=torch.float32)
    assert float(depths[1]) == 3.5
    # lattice x = 16 * 38 = 608, y = 16 * 11 = 176; 608 / 512 and 176 / 512 are exact in fp32
    P = torch.tensor([[512.0, 0, 608.0, 0], [0, 512.0, 176.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
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
    n_nan = int(np.isnan(pg).any(1).sum())
    print('output', tuple(out.shape), 'NaN outputs:', int(torch.isnan(out).sum()),
          'non-finite prev grid points:', int((~np.isfinite(pg).all(1)).sum()), 'of which NaN:', n_nan,
          'both coordinates NaN:', int(np.isnan(pg).all(1).sum()))
    assert n_nan >= H + W - 1 and int(np.isnan(pg).all(1).sum()) >= 1, 'the fixture must reach 0 / 0'
    np.savez_compressed(
        os.path.join(HERE, 'plane_sweep_nan_coords.npz'),
        cur=cur.numpy(), prev=prev.numpy(), depths=depths.numpy(), P=P.numpy(), Pinv=torch.inverse(P).numpy(),
        T=Tm.numpy(), fsf=np.float64(fsf), csf=np.float64(csf), flip=np.bool_(False),
        crop=np.asarray((0, 0), np.float64), scale=np.float64(1.0), img_shape=np.asarray((375, 1242)),
        ref_out=out.numpy(), ref_cur_grid=captured[0].numpy().reshape(-1, 2), ref_prev_grid=pg)


if __name__ == '__main__':
    main()
