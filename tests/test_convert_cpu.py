"""Offline converters (`drn_wsod_pytorch_amd.convert`) against tests/golden/convert.npz - the
outputs of the reference's own projects/WSL/tools scripts (proposal_convert.py, convert_resnet_ws_pth.py,
convert_resnet_ws_c2.py, convert_vgg.py) run unmodified on synthetic files by tests/golden/gen_golden.py - and the
round trip into the proposal loader of the data path."""
import os
import pickle

import numpy as np
import pytest
import torch

import golden_util as G
from __graft_entry__ import load_package

load_package()
from drn_wsod_pytorch_amd import convert as C  # noqa: E402
from drn_wsod_pytorch_amd import data as D  # noqa: E402

scipy_io = pytest.importorskip("scipy.io")


@pytest.fixture(scope="module")
def gold():
    return G.load("convert")


def _raw(gold):
    n = len(gold["prop_ids"])
    return [gold["raw_boxes%d" % i] for i in range(n)], [gold["raw_scores%d" % i] for i in range(n)]


def _same(got, gold, tag):
    ids = [str(x) for x in gold["prop_ids"]]
    assert [str(x) for x in got["indexes"]] == [str(x) for x in gold[tag + "_indexes"]] == ids
    for i in range(len(ids)):
        b, s = got["boxes"][i], got["scores"][i]
        assert b.dtype == gold["%s_boxes%d" % (tag, i)].dtype == np.int16
        assert s.dtype == np.float32
        assert np.array_equal(b, gold["%s_boxes%d" % (tag, i)])
        assert np.array_equal(s.reshape(-1), gold["%s_scores%d" % (tag, i)].reshape(-1))


def test_selective_search_proposals_match_reference_script(gold, tmp_path):
    raw, _ = _raw(gold)
    ids = [str(x) for x in gold["prop_ids"]]
    _same(C.proposals_from_selective_search(raw, ids), gold, "ss")
    # and from the .mat file itself (a MATLAB cell array of per-image boxes)
    cell = np.empty((len(raw),), dtype=object)
    for i, r in enumerate(raw):
        cell[i] = r
    scipy_io.savemat(str(tmp_path / "ss.mat"), {"boxes": cell})
    _same(C.proposals_from_selective_search(str(tmp_path / "ss.mat"), ids), gold, "ss")
    with pytest.raises(ValueError):
        C.proposals_from_selective_search(raw[:-1], ids)


def test_mcg_proposals_match_reference_script_and_feed_the_loader(gold, tmp_path):
    raw, scores = _raw(gold)
    ids = [str(x) for x in gold["prop_ids"]]
    for i, r, s in zip(ids, raw, scores):
        scipy_io.savemat(str(tmp_path / (i + ".mat")), {"boxes": r, "scores": s})
    got = C.proposals_from_mcg(str(tmp_path), ids)
    _same(got, gold, "mcg")
    assert got["scores"][1].shape == (1,)  # the single-proposal image stays indexable
    # the Flickr dumps' variable names, files named after the image file instead of the id
    for i, r, s in zip(ids, raw, scores):
        scipy_io.savemat(str(tmp_path / ("img_" + i + ".mat")), {"bboxes": r, "bboxes_scores": s})
    _same(C.proposals_from_mcg(str(tmp_path), ids, file_stems=["img_" + i for i in ids], flickr=True), gold, "mcg")
    # written file -> data.load_proposals_into_dataset (detectron2/data/build.py:102-153): sorted by score, descending
    path = str(tmp_path / "mcg_d.pkl")
    C.write_proposal_file(path, got)
    with open(path, "rb") as f:
        assert sorted(pickle.load(f)) == ["boxes", "indexes", "scores"]
    recs = D.load_proposals_into_dataset([{"image_id": i} for i in reversed(ids)], path)
    for r in recs:
        k = ids.index(r["image_id"])
        order = np.argsort(got["scores"][k])[::-1]
        assert np.array_equal(r["proposal_boxes"], got["boxes"][k][order])
        assert np.array_equal(r["proposal_objectness_logits"], got["scores"][k][order])


def test_checkpoint_key_renaming_matches_reference_scripts(gold, tmp_path):
    pth_in = [str(k) for k in gold["pth_in"]]
    out = C.rename_ws_pth_keys({k: i for i, k in enumerate(pth_in)})
    assert [k for k, _ in sorted(out.items(), key=lambda kv: kv[1])] == [str(k) for k in gold["pth_out"]]

    c2_in = [str(k) for k in gold["c2_in"]]
    blobs = {k: i for i, k in enumerate(c2_in) if not k.endswith("_momentum")}  # the scripts' loader drops momentum blobs
    out = C.rename_ws_c2_blobs(blobs)
    assert list(out.keys()) == [str(k) for k in gold["c2_out_keys"]]
    assert list(out.values()) == gold["c2_out_src"].tolist()

    vgg_in = [str(k) for k in gold["vgg_in"]]
    out = C.rename_vgg_blobs({k: i for i, k in enumerate(vgg_in)})
    assert list(out.keys()) == [str(k) for k in gold["vgg_out_keys"]]
    assert list(out.values()) == gold["vgg_out_src"].tolist()

    # file to file, the way the scripts are used
    src, dst = str(tmp_path / "c2.pkl"), str(tmp_path / "c2_out.pkl")
    with open(src, "wb") as f:
        pickle.dump({"blobs": {k: np.full((2,), float(i), np.float32) for i, k in enumerate(c2_in)}}, f, 2)
    C.convert_checkpoint_file(src, dst, "ws_c2")
    with open(dst, "rb") as f:
        got = pickle.load(f)
    assert list(got.keys()) == [str(k) for k in gold["c2_out_keys"]]
    assert [int(v[0]) for v in got.values()] == gold["c2_out_src"].tolist()
    src, dst = str(tmp_path / "ws.pth"), str(tmp_path / "ws_out.pth")
    torch.save({"state_dict": {k: torch.full((2,), float(i)) for i, k in enumerate(pth_in)}, "epoch": 120}, src)
    C.convert_checkpoint_file(src, dst, "ws_pth")
    got = torch.load(dst)
    assert [k for k, _ in sorted(got.items(), key=lambda kv: float(kv[1][0]))] == [str(k) for k in gold["pth_out"]]
    with pytest.raises(ValueError):
        C.convert_checkpoint_file(src, dst, "caffe")
    assert not os.path.exists(dst + ".tmp.%d" % os.getpid())
