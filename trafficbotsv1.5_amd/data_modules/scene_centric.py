"""`SceneCentricPreProcessing` (data_modules/scene_centric.py:8-165): re-keys a packed-h5 batch into the sc/*, gt/*,
ref/* tensors the model consumes (tl_mode=lane, navi_mode=dest). Pure indexing on whatever device the batch is on."""
from typing import Dict, Tuple

import torch
from torch import Tensor, nn


class SceneCentricPreProcessing(nn.Module):
    def __init__(self, time_step_current: int, tl_mode: str, navi_mode: str, dropout_p_history: float, data_size) -> None:
        super().__init__()
        if tl_mode != "lane" or navi_mode != "dest":
            raise NotImplementedError("the MI355X path implements tl_mode=lane, navi_mode=dest")
        self.n_step_hist = time_step_current + 1
        self.tl_mode, self.navi_mode, self.dropout_p_history = tl_mode, navi_mode, dropout_p_history
        self.model_kwargs = {
            "tl_mode": tl_mode, "navi_mode": navi_mode, "navi_dim": None,
            "n_mp_pl_node": data_size["map/valid"][-1], "mp_attr_dim": data_size["map/type"][-1],
            "tl_state_dim": data_size["tl_stop/state"][-1],
            "ag_motion_dim": data_size["agent/spd"][-1] + data_size["agent/acc"][-1] + data_size["agent/yaw_rate"][-1],
            "ag_attr_dim": data_size["agent/size"][-1] + data_size["agent/type"][-1],
        }

    @staticmethod
    def _merge_invalid_tl_into_state(tl_valid: Tensor, tl_state: Tensor) -> Tuple[Tensor, Tensor]:
        seen = tl_valid.any(-1)
        unknown = (~tl_valid) & seen.unsqueeze(-1)
        tl_state = tl_state.clone()
        tl_state[..., 0] |= unknown
        return seen, tl_state

    def forward(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        b, H = batch, self.n_step_hist
        p = "" if self.training else "history/"
        b["sc/mp_valid"] = b["map/valid"].clone()
        b["sc/mp_attr"] = b["map/type"].type_as(b["map/pos"])
        # (slices, not `[..., [1]]`: a Python list index becomes a host tensor whose copy to the device blocks the host until the stream has
        #  drained - once per training step, in front of the next step's replay: tools/train_host_timeline.py)
        #  (contiguous copies of the slices, as the list index made them: on the CPU atan2 takes another - vectorised - path on contiguous
        #  operands and the oracle comparison of tests/test_abi_and_host.py is bit-exact)
        b["sc/mp_pose"] = torch.cat([b["map/pos"][..., :2], torch.atan2(b["map/dir"][..., 1:2].contiguous(), b["map/dir"][..., 0:1].contiguous())], -1)
        b["sc/tl_valid"], b["sc/tl_state"] = self._merge_invalid_tl_into_state(b[p + "tl_lane/valid"][:, :, :H],
                                                                               b[p + "tl_lane/state"][:, :, :H])
        b["sc/tl_attr"] = b[p + "tl_lane/idx"]
        n_sc = b["sc/mp_pose"].shape[0]
        b["sc/tl_pose"] = b["sc/mp_pose"][torch.arange(n_sc, device=b["sc/mp_pose"].device).unsqueeze(1), b["sc/tl_attr"], 0]
        b["sc/ag_valid"] = b[p + "agent/valid"][:, :, :H].clone()
        b["sc/ag_attr"] = torch.cat([b[p + "agent/size"], b[p + "agent/type"].type_as(b[p + "agent/size"])], -1)
        b["sc/ag_motion"] = torch.cat([b[p + "agent/" + k][:, :, :H] for k in ("spd", "acc", "yaw_rate")], -1)
        b["sc/ag_pose"] = torch.cat([b[p + "agent/pos"][:, :, :H, :2], b[p + "agent/yaw_bbox"][:, :, :H]], -1)
        if "agent/valid" in b:
            b["gt/ag_valid"] = b["agent/valid"]
            b["gt/ag_motion"] = torch.cat([b["agent/spd"], b["agent/acc"], b["agent/yaw_rate"]], -1)
            b["gt/ag_pose"] = torch.cat([b["agent/pos"][..., :2], b["agent/yaw_bbox"]], -1)
            b["gt/ag_navi"] = b["agent/dest"]
            b["gt/tl_valid"], b["gt/tl_state"] = self._merge_invalid_tl_into_state(b["tl_lane/valid"], b["tl_lane/state"])
        for k in ("type", "role", "size"):
            b["ref/ag_" + k] = b[p + "agent/" + k]
        b["ref/mp_type"] = b["map/type"]
        if self.training and 0 < self.dropout_p_history <= 1.0:
            keep = 1 - self.dropout_p_history
            b["sc/mp_valid"][:, :, 1:] &= torch.bernoulli(torch.full_like(b["sc/mp_valid"][:, :, 1:], keep, dtype=torch.float32)).bool()
            b["sc/ag_valid"][..., :-1] &= torch.bernoulli(torch.full_like(b["sc/ag_valid"][..., :-1], keep, dtype=torch.float32)).bool()
        return b
