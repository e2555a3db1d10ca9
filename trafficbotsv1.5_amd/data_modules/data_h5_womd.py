"""`DataH5womd` / `DatasetTrain` / `DatasetVal` (data_modules/data_h5_womd.py:11-231) on a fixed-shape scene pack.

The reference keeps one gzip-4 + shuffle compressed h5 group per episode (scripts/pack_h5_womd.py:335-342) and, on every
`__getitem__`, opens the file, decompresses ~30 datasets and down-casts float32 to float16
(data_h5_womd.py:36-43). SURVEY.md §8f row 4 asks for the layout that feeds the GPU instead: the episodes here live in ONE
uncompressed structure-of-arrays file - per key a contiguous `[n_episode, *shape]` array, 4 KiB aligned, already in the dtype
the reference's loader delivers (float32 -> float16 at pack time: same values, half the bytes) - so an episode is ~30
memcpy-able slices of a memory map and a batch of consecutive episodes is ONE contiguous slice per key, copied to the device
without touching the elements (`ScenePack.batch`). h5py is not needed to read a pack; `convert_h5` (needs h5py, run wherever
the h5 files live) writes one from the reference's files, `write_scene_pack` from any iterable of episode dicts.

File: b"TBXPACK1" | u64 header length | JSON header {n, keys: [{name, dtype, shape, offset}], attrs: {name: [per-episode values]}}
| padding to 4 KiB | key arrays. bool is stored as uint8 and viewed back as bool.
"""
import json
from pathlib import Path
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

try:  # Lightning is optional (absent on the GPU box): the module keeps the LightningDataModule surface either way
    from pytorch_lightning import LightningDataModule
except ImportError:  # pragma: no cover
    class LightningDataModule:  # type: ignore
        def __init__(self) -> None:
            pass

MAGIC = b"TBXPACK1"
ALIGN = 4096


def loader_dtype(dt: np.dtype) -> np.dtype:
    """What the reference's `__getitem__` hands out for a stored dtype (data_h5_womd.py:41-42): float32 -> float16."""
    return np.dtype(np.float16) if np.dtype(dt) == np.dtype("<f4") else np.dtype(dt)


def _align(x: int) -> int:
    return (x + ALIGN - 1) // ALIGN * ALIGN


def write_scene_pack(path: str, episodes: Iterable[Dict[str, Any]], tensor_size: Dict[str, Tuple[int, ...]],
                     attr_keys: Sequence[str] = (), n: Optional[int] = None) -> int:
    """Writes episodes (dicts of arrays with the shapes of `tensor_size`; extra `attr_keys` entries are kept per episode) as a
    scene pack, STREAMING: `episodes` is consumed once, one episode in memory at a time (a WOMD training split is ~487 k
    episodes - data_h5_womd.py / scripts/pack_h5_womd.py:194). `n` = number of episodes (taken from len(episodes) when it has
    one): the layout - per key one [n, *shape] array at a 4 KiB-aligned offset - is fixed before the first episode is read.
    Per-episode attributes go to a length-prefixed JSON footer behind the arrays. Returns the number of episodes."""
    if n is None:
        n = len(episodes)  # a generator has none: pass n (convert_h5 reads it from the h5 attribute `data_len`)
    assert n > 0
    it = iter(episodes)
    first = next(it)
    keys = []
    for k, shape in tensor_size.items():
        dt = loader_dtype(np.asarray(first[k]).dtype)
        keys.append({"name": k, "dtype": "bool" if dt == np.bool_ else dt.str, "shape": list(shape)})
    # two passes over the header: offsets depend on its length
    off = 0
    for _ in range(3):
        hdr = json.dumps({"n": n, "keys": keys, "attrs_offset": off}).encode()
        off = _align(len(MAGIC) + 8 + len(hdr) + 32)  # 32 B of slack: the footer offset's digits may still grow
        for kd in keys:
            kd["offset"] = off
            kd["_item"] = int(np.prod(kd["shape"], dtype=np.int64)) * np.dtype(np.uint8 if kd["dtype"] == "bool" else kd["dtype"]).itemsize
            off = _align(off + n * kd["_item"])
    items = {kd["name"]: kd.pop("_item") for kd in keys}
    hdr = json.dumps({"n": n, "keys": keys, "attrs_offset": off}).encode()
    assert len(MAGIC) + 8 + len(hdr) <= keys[0]["offset"]
    attrs = {k: [] for k in attr_keys}
    count = 0
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(np.uint64(len(hdr)).tobytes())
        f.write(hdr)
        f.truncate(off)
        e, first = first, None
        while e is not None:
            assert count < n, "more episodes than announced"
            for kd in keys:
                a = np.asarray(e[kd["name"]])
                assert tuple(a.shape) == tuple(kd["shape"]), (kd["name"], a.shape, kd["shape"])
                f.seek(kd["offset"] + count * items[kd["name"]])
                f.write(np.ascontiguousarray(a, dtype=np.uint8 if kd["dtype"] == "bool" else np.dtype(kd["dtype"])).tobytes())
            for k in attr_keys:
                attrs[k].append(e[k] if isinstance(e[k], str) else (e[k].decode() if isinstance(e[k], bytes) else np.asarray(e[k]).tolist()))
            count += 1
            e = next(it, None)
        assert count == n, f"{count} episodes written, {n} announced"
        foot = json.dumps(attrs).encode()
        f.seek(off)
        f.write(np.uint64(len(foot)).tobytes())
        f.write(foot)
    return n


def convert_h5(h5_path: str, out_path: str, tensor_size: Dict[str, Tuple[int, ...]], with_attrs: bool = False) -> int:
    """A reference h5 file (scripts/pack_h5_womd.py layout: one group per episode index) -> scene pack. Needs h5py. One episode
    is resident at a time (the generator below is consumed once by write_scene_pack)."""
    import h5py  # noqa: WPS433 - only where the h5 files live

    attr_keys = ("scenario_id", "scenario_center", "scenario_yaw", "with_map") if with_attrs else ()
    with h5py.File(h5_path, "r", libver="latest", swmr=True) as hf:
        n = int(hf.attrs["data_len"])

        def episodes():
            for i in range(n):
                g = hf[str(i)]
                e = {k: np.asarray(g[k]) for k in tensor_size}
                for a in attr_keys:
                    e[a] = g.attrs[a]
                yield e

        return write_scene_pack(out_path, episodes(), tensor_size, attr_keys, n=n)


class ScenePack:
    """Memory-mapped reader of one pack file."""

    def __init__(self, path: str) -> None:
        self.path = str(path)
        with open(self.path, "rb") as f:
            if f.read(len(MAGIC)) != MAGIC:
                raise ValueError(f"{path}: not a scene pack")
            hlen = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
            hdr = json.loads(f.read(hlen).decode())
        self.n, self.attrs = int(hdr["n"]), hdr.get("attrs", {})
        if "attrs_offset" in hdr:  # per-episode attributes: length-prefixed JSON footer behind the arrays
            with open(self.path, "rb") as f:
                f.seek(int(hdr["attrs_offset"]))
                flen = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
                self.attrs = json.loads(f.read(flen).decode())
        self.keys = {kd["name"]: kd for kd in hdr["keys"]}
        self._maps: Dict[str, np.ndarray] = {}

    def array(self, key: str) -> np.ndarray:
        """[n, *shape] view of a key (bool keys as bool)."""
        if key not in self._maps:
            kd = self.keys[key]
            is_bool = kd["dtype"] == "bool"
            m = np.memmap(self.path, mode="r", dtype=np.uint8 if is_bool else np.dtype(kd["dtype"]), offset=kd["offset"],
                          shape=(self.n, *kd["shape"]))
            self._maps[key] = m.view(np.bool_) if is_bool else m
        return self._maps[key]

    def episode(self, idx: int, keys: Optional[Iterable[str]] = None) -> Dict[str, np.ndarray]:
        return {k: np.array(self.array(k)[idx], copy=True) for k in (keys or self.keys)}  # writable copies, as the reference's items

    def batch(self, start: int, size: int, device=None, keys: Optional[Iterable[str]] = None) -> Dict[str, torch.Tensor]:
        """Episodes [start, start + size) as batched tensors: one contiguous slice per key, no per-episode collation.
        On a device: through pinned staging buffers, copies enqueued without waiting for each other."""
        out = {}
        for k in (keys or self.keys):
            t = torch.from_numpy(np.array(self.array(k)[start:start + size], copy=True))  # one memcpy of a contiguous slice
            if device is not None and torch.device(device).type == "cuda":
                t = t.pin_memory().to(device, non_blocking=True)
            out[k] = t
        out["episode_idx"] = torch.arange(start, start + size)
        return out

    def __getstate__(self):  # DataLoader workers re-open their own maps
        return {"path": self.path}

    def __setstate__(self, st):
        self.__init__(st["path"])


class DatasetBase(Dataset):
    def __init__(self, h5_filepath: str, tensor_size: Dict[str, Tuple], scenario_dir: Optional[str] = None) -> None:
        """`h5_filepath`: the reference's argument name; a `.h5` path is served from the `.tbxpack` next to it."""
        super().__init__()
        self.tensor_size = tensor_size
        p = Path(h5_filepath)
        pack = p if p.suffix == ".tbxpack" else p.with_suffix(".tbxpack")
        if not pack.exists():
            raise FileNotFoundError(f"{pack} not found: convert {p.name} once with data_modules.data_h5_womd.convert_h5")
        self.h5_filepath = str(pack)
        self.pack = ScenePack(str(pack))
        self.dataset_len = self.pack.n
        self.scenario_dir = Path(scenario_dir) if scenario_dir is not None else None
        if self.scenario_dir is not None:
            assert len(list(self.scenario_dir.glob("*"))) == self.dataset_len

    def __len__(self) -> int:
        return self.dataset_len


class DatasetTrain(DatasetBase):
    def __getitem__(self, idx: int) -> Dict[str, np.ndarray]:
        out = {"episode_idx": idx}
        out.update(self.pack.episode(idx, self.tensor_size.keys()))
        return out


class DatasetVal(DatasetBase):
    def __getitem__(self, idx: int) -> Dict[str, Any]:
        out: Dict[str, Any] = {"episode_idx": idx}
        for a in ("scenario_id", "scenario_center", "scenario_yaw", "with_map"):
            if a in self.pack.attrs:
                v = self.pack.attrs[a][idx]
                out[a] = v if isinstance(v, (str, bool)) else np.asarray(v)
        for k, _size in self.tensor_size.items():
            out[k] = np.array(self.pack.array(k)[idx], copy=True)
            if out[k].shape != tuple(_size):  # dummy agents for scalability tests (data_h5_womd.py:59-61)
                assert "agent" in k
                out[k] = np.ones(_size, dtype=out[k].dtype)
        if self.scenario_dir is not None:
            import pickle

            with open(self.scenario_dir / f"{idx}.pickle", "rb") as handle:
                out["scenario_bytes"] = pickle.load(handle).hex()
        return out


def womd_tensor_sizes(n_ag_sim: int = 64) -> Tuple[Dict[str, Tuple], Dict[str, Tuple]]:
    """(tensor_size_train, tensor_size_test) of data_h5_womd.py:95-183."""
    sd, n_mp_type, n_tl_state = 3, 11, 5
    n_ag_type, n_ag_role, ag_size_dim, n_ag_cmd = 3, 3, 3, 8
    T, Th, n_no_sim, n_mp, N, n_tl_lane, n_tl_stop = 91, 11, 256, 1024, 20, 128, 50
    A = n_ag_sim
    train = {
        "agent/valid": (A, T), "agent/pos": (A, T, sd), "agent/vel": (A, T, 2), "agent/spd": (A, T, 1), "agent/acc": (A, T, 1),
        "agent/yaw_bbox": (A, T, 1), "agent/yaw_rate": (A, T, 1), "agent/type": (A, n_ag_type), "agent/cmd": (A, n_ag_cmd),
        "agent/role": (A, n_ag_role), "agent/size": (A, ag_size_dim), "agent/goal": (A, 4), "agent/dest": (A,),
        "map/valid": (n_mp, N), "map/type": (n_mp, n_mp_type), "map/pos": (n_mp, N, sd), "map/dir": (n_mp, N, sd), "map/boundary": (4,),
        "tl_lane/valid": (n_tl_lane, T), "tl_lane/state": (n_tl_lane, T, n_tl_state), "tl_lane/idx": (n_tl_lane,),
        "tl_stop/valid": (n_tl_stop, T), "tl_stop/state": (n_tl_stop, T, n_tl_state), "tl_stop/pos": (n_tl_stop, sd), "tl_stop/dir": (n_tl_stop, sd),
    }
    test = {"history/agent/object_id": (A,), "history/agent_no_sim/object_id": (n_no_sim,)}
    for k in ("valid", "pos", "vel", "spd", "acc", "yaw_bbox", "yaw_rate"):
        test[f"history/agent/{k}"] = (A, Th) + train[f"agent/{k}"][2:]
    for k in ("type", "role", "size"):
        test[f"history/agent/{k}"] = train[f"agent/{k}"]
    for k, tail in (("valid", ()), ("pos", (sd,)), ("vel", (2,)), ("spd", (1,)), ("yaw_bbox", (1,))):
        test[f"history/agent_no_sim/{k}"] = (n_no_sim, Th) + tail
    test["history/agent_no_sim/type"], test["history/agent_no_sim/size"] = (n_no_sim, n_ag_type), (n_no_sim, ag_size_dim)
    for k in ("map/valid", "map/type", "map/pos", "map/dir", "map/boundary"):
        test[k] = train[k]
    test.update({"history/tl_lane/valid": (n_tl_lane, Th), "history/tl_lane/state": (n_tl_lane, Th, n_tl_state), "history/tl_lane/idx": (n_tl_lane,),
                 "history/tl_stop/valid": (n_tl_stop, Th), "history/tl_stop/state": (n_tl_stop, Th, n_tl_state),
                 "history/tl_stop/pos": (n_tl_stop, sd), "history/tl_stop/dir": (n_tl_stop, sd)})
    return train, test


class DataH5womd(LightningDataModule):
    """Constructor and loader methods of data_h5_womd.py:72-231; `{data_dir}/{filename}.h5` is served from the scene pack
    `{data_dir}/{filename}.tbxpack`."""

    def __init__(self, data_dir: str, val_scenarios_dir: Optional[str] = None, filename_train: str = "training",
                 filename_val: str = "validation", filename_test: str = "testing", batch_size_train: int = 3, batch_size_test: int = 3,
                 num_workers: int = 4, n_ag_sim: int = 64) -> None:
        super().__init__()
        self.val_scenarios_dir = val_scenarios_dir
        self.path_train_h5 = f"{data_dir}/{filename_train}.h5"
        self.path_val_h5 = f"{data_dir}/{filename_val}.h5"
        self.path_test_h5 = f"{data_dir}/{filename_test}.h5"
        self.batch_size_train, self.batch_size_test, self.num_workers = batch_size_train, batch_size_test, num_workers
        self.tensor_size_train, self.tensor_size_test = womd_tensor_sizes(n_ag_sim)
        self.tensor_size_val = {**self.tensor_size_train, **self.tensor_size_test}

    def setup(self, stage: Optional[str] = None) -> None:
        if stage == "fit" or stage is None:
            self.train_dataset = DatasetTrain(self.path_train_h5, self.tensor_size_train)
            self.val_dataset = DatasetVal(self.path_val_h5, self.tensor_size_val, self.val_scenarios_dir)
        elif stage == "validate":
            self.val_dataset = DatasetVal(self.path_val_h5, self.tensor_size_val, self.val_scenarios_dir)
        elif stage == "test":
            self.test_dataset = DatasetVal(self.path_test_h5, self.tensor_size_test)

    def train_dataloader(self) -> DataLoader:
        return self._get_dataloader(self.train_dataset, self.batch_size_train, self.num_workers, shuffle=True)

    def val_dataloader(self) -> DataLoader:
        return self._get_dataloader(self.val_dataset, self.batch_size_test, self.num_workers, shuffle=False)

    def test_dataloader(self) -> DataLoader:
        return self._get_dataloader(self.test_dataset, self.batch_size_test, self.num_workers, shuffle=False)

    @staticmethod
    def _get_dataloader(ds: Dataset, batch_size: int, num_workers: int, shuffle: bool) -> DataLoader:
        """One process per GPU: with an initialised process group every rank draws from its own 1/N share of the episodes
        (a DistributedSampler, which Lightning's ddp strategy injected in the reference, run.py:50-52 - call
        `loader.sampler.set_epoch(epoch)` per epoch for a fresh shuffle); single process: the reference's plain loader."""
        import torch.distributed as dist

        sampler = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.utils.data.distributed import DistributedSampler

            sampler = DistributedSampler(ds, num_replicas=dist.get_world_size(), rank=dist.get_rank(), shuffle=shuffle, drop_last=False)
        return DataLoader(ds, batch_size=batch_size, num_workers=num_workers, pin_memory=torch.cuda.is_available(),
                          shuffle=shuffle if sampler is None else False, sampler=sampler, drop_last=False,
                          persistent_workers=num_workers > 0)
