"""Writes tests/golden/womd_tensor_sizes.json: `tensor_size_train / _test / _val` of the REFERENCE's DataH5womd
(src/data_modules/data_h5_womd.py:72-183), imported in the build container behind stub modules for pytorch_lightning / h5py
(absent here; only class definitions need them). Run: python tests/golden/make_tensor_sizes.py"""
import json
import sys
import types
from pathlib import Path

pl = types.ModuleType("pytorch_lightning")
pl.LightningDataModule = type("LightningDataModule", (), {"__init__": lambda self: None})
sys.modules["pytorch_lightning"] = pl
sys.modules["h5py"] = types.ModuleType("h5py")
sys.path.insert(0, "/root/reference/src")
from data_modules.data_h5_womd import DataH5womd  # noqa: E402

out = {}
for n_ag in (64, 128):
    dm = DataH5womd(data_dir="/nonexistent", n_ag_sim=n_ag)
    out[str(n_ag)] = {name: {k: list(v) for k, v in getattr(dm, name).items()} for name in ("tensor_size_train", "tensor_size_test", "tensor_size_val")}
Path(__file__).with_name("womd_tensor_sizes.json").write_text(json.dumps(out, indent=0, sort_keys=True))
print({k: {n: len(v) for n, v in d.items()} for k, d in out.items()})
