"""`get_rel_pose` / `get_rel_dist` / `get_tgt_knn_idx` with the reference's signatures (utils/rpe.py:8-90), over the HIP entry
points. The hot path itself never materialises [n_sc, n_src, n_tgt, *] tensors: `tbx_knn_embed` forms the relative poses, ranks
and gathers in registers. So `get_rel_pose` / `get_rel_dist` return LAZY results (`PairGeometry` views that remember the poses
they came from) and `get_tgt_knn_idx` hands those poses to the fused search; the dense tensors are produced on demand by
`tbx_rel_pose_dense` (`.dense()`, or any tensor attribute / operation through `__torch_function__`-free explicit access), with the
same expressions as the search - a dense distance equals the key the search ranks by, bit for bit."""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from .. import hip


class PairGeometry:
    """The (source, target) poses of one `get_rel_pose` / `get_rel_dist` call, and which of its two results this object stands for
    ("pose": [n_sc, n_src, n_tgt, 3], "dist": [n_sc, n_src, n_tgt]). `dense()` materialises it (cached)."""

    def __init__(self, kind: str, src_pose: Tensor, src_invalid: Tensor, tgt_pose: Tensor, tgt_invalid: Tensor, rotate: bool):
        self.kind, self.rotate = kind, rotate
        self.src_pose, self.src_invalid, self.tgt_pose, self.tgt_invalid = src_pose, src_invalid, tgt_pose, tgt_invalid
        self._dense = None

    @property
    def shape(self):
        n, S, _ = self.src_pose.shape
        T = self.tgt_pose.shape[1]
        return torch.Size((n, S, T, 3) if self.kind == "pose" else (n, S, T))

    def dense(self) -> Tensor:
        if self._dense is None:
            rel, dist = hip.rel_pose_dense(self.src_pose, self.src_invalid, self.tgt_pose, self.tgt_invalid,
                                           want_rel_pose=self.kind == "pose", want_dist=self.kind == "dist")
            self._dense = rel if self.kind == "pose" else dist
        return self._dense

    def __getitem__(self, item):
        return self.dense()[item]

    def __getattr__(self, name):  # anything else a tensor offers (.amin, .isinf, ...): on the dense tensor
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.dense(), name)


def _pose3(x: Tensor) -> Tensor:
    return x.float().contiguous()


def _u8(m: Tensor) -> Tensor:
    return m.to(torch.uint8).contiguous()


@torch.no_grad()
def get_rel_pose(pose: Tensor, invalid: Tensor, pose2: Optional[Tensor] = None, invalid2: Optional[Tensor] = None
                 ) -> Tuple[PairGeometry, PairGeometry]:
    """utils/rpe.py:8-37. pose [n_sc, n_src, 3] (x, y, yaw), invalid [n_sc, n_src]; pose2 / invalid2 likewise or None (= the sources).
    -> rel_pose [n_sc, n_src, n_tgt, 3], rel_dist [n_sc, n_src, n_tgt] (+inf on invalid pairs), both lazy."""
    if pose2 is None:
        pose2, invalid2 = pose, invalid
    a = (_pose3(pose), _u8(invalid), _pose3(pose2), _u8(invalid2))
    return PairGeometry("pose", *a, rotate=True), PairGeometry("dist", *a, rotate=True)


@torch.no_grad()
def get_rel_dist(xy: Tensor, invalid: Tensor, xy2: Optional[Tensor] = None, invalid2: Optional[Tensor] = None) -> PairGeometry:
    """utils/rpe.py:41-58: distances of un-rotated positions [n_sc, n_src, 2] (a zero yaw makes the rotation the identity, exactly)."""
    if xy2 is None:
        xy2, invalid2 = xy, invalid
    z = lambda p: torch.cat([p.float(), torch.zeros_like(p[..., :1], dtype=torch.float32)], -1).contiguous()
    return PairGeometry("dist", z(xy), _u8(invalid), z(xy2), _u8(invalid2), rotate=False)


@torch.no_grad()
def get_tgt_knn_idx(tgt_invalid: Tensor, rel_pose: Optional[PairGeometry], rel_dist: PairGeometry, n_tgt_knn: int,
                    dist_limit: Union[float, Tensor]) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """utils/rpe.py:61-90 -> idx_tgt int64 [n_sc, n_src, K], tgt_invalid_knn bool [n_sc, n_src, K], rpe [n_sc, n_src, K, 3] | None.
    The K smallest distances as a SET (the reference's topk(sorted=False) leaves the order open; ties go to the lower index);
    slots whose distance is +inf or beyond dist_limit are flagged invalid. rel_dist / rel_pose must come from this module's
    get_rel_pose / get_rel_dist: the search runs on the poses behind them (tbx_knn_embed), not on a dense matrix."""
    if not isinstance(rel_dist, PairGeometry):
        raise TypeError("get_tgt_knn_idx: rel_dist must be the result of utils.rpe.get_rel_pose / get_rel_dist (the fused K-nearest "
                        "search ranks relative poses it forms itself; a dense distance matrix is not an input of the HIP path)")
    if torch.is_tensor(dist_limit):
        raise NotImplementedError("per-target dist_limit tensors are not used by any default entry point")
    g = rel_dist
    n_tgt = g.tgt_pose.shape[1]
    assert 0 < n_tgt_knn < n_tgt  # (utils/rpe.py:79)
    # tgt_invalid is the mask the distances were built with (every call site passes the same one, e.g. agent_encoder.py:340-352)
    idx, inv, rel, _ = hip.knn_embed(g.src_pose, g.src_invalid, g.tgt_pose, _u8(tgt_invalid) | g.tgt_invalid, n_tgt_knn, float(dist_limit),
                                     want_rel_pose=rel_pose is not None, want_emb=False)
    return idx.long(), inv.bool(), rel
