"""the data-loading boundary (`data_modules/data_h5_womd.py`) on the fixed-shape scene pack. CPU tests:
the module's tensor-size tables equal the reference's (golden fixture generated from the reference's class), a pack round-trips
to exactly what the reference's `__getitem__` hands out (float32 -> float16, everything else untouched: data_h5_womd.py:36-43),
a batch of consecutive episodes equals the default collation of the per-episode dicts, and the pre-processing accepts it."""
import json
from importlib import import_module

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def D(tb):
    return import_module("trafficbots_amd.data_modules.data_h5_womd")


def test_tensor_sizes_equal_the_references(D, golden_dir):
    g = json.loads((golden_dir / "womd_tensor_sizes.json").read_text())
    for n_ag in (64, 128):
        dm = D.DataH5womd(data_dir="/nonexistent", n_ag_sim=n_ag)
        for name in ("tensor_size_train", "tensor_size_test", "tensor_size_val"):
            ours = {k: list(v) for k, v in getattr(dm, name).items()}
            assert ours == g[str(n_ag)][name], name
            assert list(ours) == list(g[str(n_ag)][name]) or sorted(ours) == sorted(g[str(n_ag)][name])


def _episodes(tb, n, n_ag=16, n_mp=64, n_tl=8):
    eps = []
    for s in range(n):
        b = tb.synthetic.make_scene(1, n_ag, n_mp, n_tl, seed=100 + s)
        eps.append({k: v[0].numpy() for k, v in b.items()})
    return eps


def test_pack_round_trip_equals_reference_loader_semantics(tb, D, tmp_path):
    eps = _episodes(tb, 5)
    sizes = {k: tuple(v.shape) for k, v in eps[0].items()}
    path = tmp_path / "training.tbxpack"
    assert D.write_scene_pack(str(path), eps, sizes) == 5
    ds = D.DatasetTrain(str(tmp_path / "training.h5"), sizes)  # the reference's path argument: served from the pack next to it
    assert len(ds) == 5
    saw_half = False
    for i in (0, 3, 4):
        item = ds[i]
        assert item["episode_idx"] == i
        for k, x in eps[i].items():
            want = np.ascontiguousarray(x, dtype=np.float16 if x.dtype == np.dtype("<f4") else None)  # data_h5_womd.py:41-42
            assert item[k].dtype == want.dtype and item[k].shape == want.shape, k
            assert np.array_equal(item[k], want), k
            assert item[k].flags["C_CONTIGUOUS"]
            saw_half = saw_half or want.dtype == np.float16
    assert saw_half
    pack = D.ScenePack(str(path))
    for kd in pack.keys.values():
        assert kd["offset"] % D.ALIGN == 0
    # a batch of consecutive episodes: one contiguous slice per key == default collation of the items
    from torch.utils.data import default_collate

    want = default_collate([ds[i] for i in (1, 2, 3)])
    got = pack.batch(1, 3)
    assert sorted(got) == sorted(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # DataLoader path (workers re-open the map), shuffled and not
    dl = D.DataH5womd._get_dataloader(ds, 2, 0, shuffle=False)
    first = next(iter(dl))
    assert torch.equal(first["agent/pos"], want["agent/pos"][:0].new_tensor(np.stack([ds[0]["agent/pos"], ds[1]["agent/pos"]])))


def test_val_items_attrs_and_dummy_agents(tb, D, tmp_path):
    eps = _episodes(tb, 3)
    for i, e in enumerate(eps):
        e.update(scenario_id=f"scn{i:03d}", scenario_center=np.array([1.0 * i, 2.0, 0.0]), scenario_yaw=0.1 * i, with_map=bool(i % 2))
    sizes = {k: tuple(v.shape) for k, v in eps[0].items() if "/" in k}
    D.write_scene_pack(str(tmp_path / "validation.tbxpack"), eps, sizes, attr_keys=("scenario_id", "scenario_center", "scenario_yaw", "with_map"))
    big = dict(sizes)
    big["agent/valid"] = (32,) + sizes["agent/valid"][1:]  # more agents than packed: dummy ones, as the reference does
    ds = D.DatasetVal(str(tmp_path / "validation.h5"), big)
    it = ds[2]
    assert it["scenario_id"] == "scn002" and it["with_map"] is False and np.allclose(it["scenario_center"], [2.0, 2.0, 0.0])
    assert it["agent/valid"].shape == (32,) + sizes["agent/valid"][1:] and it["agent/valid"].all()
    assert np.array_equal(it["map/pos"], np.asarray(eps[2]["map/pos"], dtype=np.float16))
    with pytest.raises(FileNotFoundError):
        D.DatasetTrain(str(tmp_path / "missing.h5"), sizes)
    (tmp_path / "bad.tbxpack").write_bytes(b"not a pack at all")
    with pytest.raises(ValueError):
        D.ScenePack(str(tmp_path / "bad.tbxpack"))


def test_preprocessing_accepts_a_pack_batch(tb, D, tmp_path):
    """The batch a pack delivers (float16 / bool / int64) goes through SceneCentricPreProcessing like the reference's batches."""
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    eps = _episodes(tb, 2, n_ag=8, n_mp=64, n_tl=8)
    sizes = {k: tuple(v.shape) for k, v in eps[0].items()}
    D.write_scene_pack(str(tmp_path / "training.tbxpack"), eps, sizes)
    batch = D.ScenePack(str(tmp_path / "training.tbxpack")).batch(0, 2)
    ref = {k: torch.from_numpy(np.stack([e[k] for e in eps])) for k in sizes}
    scfg = tb.config.default_sim_cfg()
    scfg["pre_processing"]["scene_centric"]["dropout_p_history"] = -1.0
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=4), data_size=tb.synthetic.DATA_SIZE, **scfg)
    with torch.no_grad():
        a = wm.pre_processing({k: v.clone() for k, v in batch.items()})
        b = wm.pre_processing({k: v.clone() for k, v in ref.items()})
    for k in ("sc/ag_pose", "sc/mp_pose", "gt/ag_pose"):
        assert a[k].shape == b[k].shape
        torch.testing.assert_close(a[k].float(), b[k].float(), rtol=2e-3, atol=0.15)  # float16 storage of coordinates up to +-150 m
    assert torch.equal(a["sc/ag_valid"], b["sc/ag_valid"])


def test_convert_h5_through_a_stand_in_h5py(tb, D, tmp_path, monkeypatch):
    """convert_h5 reads the reference's file layout (attrs['data_len'], one group per episode index, per-group attrs) through
    the h5py API. h5py is absent here: a minimal in-memory stand-in of the few calls used drives the same code."""
    import sys
    import types

    eps = _episodes(tb, 3)
    for i, e in enumerate(eps):
        e["_attrs"] = dict(scenario_id=f"s{i}", scenario_center=np.array([float(i), 0.0, 0.0]), scenario_yaw=0.5 * i, with_map=True)

    class Group(dict):
        def __init__(self, arrays, attrs):
            super().__init__(arrays)
            self.attrs = attrs

    class File:
        def __init__(self, path, mode="r", **kw):
            self.attrs = {"data_len": len(eps)}
            self._g = {str(i): Group({k: v for k, v in e.items() if k != "_attrs"}, e["_attrs"]) for i, e in enumerate(eps)}

        def __getitem__(self, k):
            return self._g[k]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    fake = types.ModuleType("h5py")
    fake.File = File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    sizes = {k: tuple(v.shape) for k, v in eps[0].items() if k != "_attrs"}
    out = tmp_path / "validation.tbxpack"
    assert D.convert_h5("validation.h5", str(out), sizes, with_attrs=True) == 3
    ds = D.DatasetVal(str(out), sizes)
    it = ds[1]
    assert it["scenario_id"] == "s1" and it["with_map"] is True and float(it["scenario_yaw"]) == 0.5
    assert np.array_equal(it["agent/pos"], np.asarray(eps[1]["agent/pos"], dtype=np.float16))


def test_write_scene_pack_streams_a_generator(tb, D, tmp_path):
    """The writer consumes its episodes ONCE and holds one at a time (a WOMD split is ~487 k episodes: the ingestion path may
    not collect them): a generator that frees every episode it handed out before producing the next one must give the same file
    as the list, and a generator without an announced length is refused."""
    eps = _episodes(tb, 4)
    sizes = {k: tuple(v.shape) for k, v in eps[0].items()}
    live = {"n": 0, "max": 0}

    class Ep(dict):
        def __init__(self, d):
            super().__init__(d)
            live["n"] += 1
            live["max"] = max(live["max"], live["n"])

        def __del__(self):
            live["n"] -= 1

    def gen():
        for e in eps:
            yield Ep(e)

    a, b = tmp_path / "a.tbxpack", tmp_path / "b.tbxpack"
    assert D.write_scene_pack(str(a), eps, sizes) == 4
    assert D.write_scene_pack(str(b), gen(), sizes, n=4) == 4
    assert a.read_bytes() == b.read_bytes()
    assert live["max"] <= 2  # the one being written + the one just fetched
    with pytest.raises(TypeError):
        D.write_scene_pack(str(b), gen(), sizes)
    with pytest.raises(AssertionError):
        D.write_scene_pack(str(b), gen(), sizes, n=5)


@pytest.mark.gpu
def test_scene_pack_batch_on_the_device_feeds_the_hot_path(tb, D, tmp_path):
    """`ScenePack.batch(start, size, device)` - one contiguous slice per key through pinned staging - lands on the GPU with the
    values, dtypes and shapes the default collation of the per-episode items would have, and the HIP hot path accepts it: the
    pre-processing + the map encoder run on the batch and agree with the same scenes handed over as fp32 synthetic tensors (the
    pack stores what the reference's loader delivers: float32 -> float16, data_h5_womd.py:41-42, so inputs agree to fp16)."""
    dev = torch.device("cuda:0")
    eps = _episodes(tb, 3, n_ag=8, n_mp=64, n_tl=8)
    sizes = {k: tuple(v.shape) for k, v in eps[0].items()}
    path = tmp_path / "training.tbxpack"
    D.write_scene_pack(str(path), eps, sizes)
    pack = D.ScenePack(str(path))
    b = pack.batch(1, 2, device=dev)
    assert b["episode_idx"].tolist() == [1, 2]
    for k in sizes:
        v = b[k]
        assert v.is_cuda and v.shape[0] == 2
        want = np.stack([np.ascontiguousarray(e[k], dtype=np.float16 if e[k].dtype == np.dtype("<f4") else None) for e in eps[1:3]])
        assert v.dtype == torch.from_numpy(want).dtype and tuple(v.shape) == want.shape, k
        assert np.array_equal(v.cpu().numpy(), want), k
    W = import_module("trafficbots_amd.pl_modules.waymo_motion")
    wm = W.WaymoMotion(model=tb.config.default_model_cfg(n_tgt_knn=4), data_size=tb.synthetic.DATA_SIZE, **tb.config.default_sim_cfg())
    tb.utils.det_fill(wm.model, 0)
    wm = wm.to(dev).eval()
    ref = {k: torch.from_numpy(np.stack([e[k] for e in eps[1:3]])).to(dev) for k in eps[0]}
    outs = []
    for batch in (b, ref):
        full = {k: (v.float() if v.dtype == torch.float16 else v) for k, v in batch.items() if k in sizes}
        bb = wm.pre_processing({**full, **tb.synthetic.to_history_batch(full)})  # eval mode reads the `history/*` view
        outs.append(wm.model.mp_encoder(bb["sc/mp_valid"], bb["sc/mp_attr"], bb["sc/mp_pose"], bb["ref/mp_type"]))
    assert torch.equal(outs[0]["mp_token_invalid"], outs[1]["mp_token_invalid"])
    torch.testing.assert_close(outs[0]["mp_token_pose"], outs[1]["mp_token_pose"], rtol=2e-3, atol=0.1)  # fp16 positions: 0.06 m at 150 m
    assert torch.isfinite(outs[0]["mp_token_feature"]).all()
