#!/usr/bin/env python
"""Generates tests/golden/reconstruct_calls.npz by RUNNING the reference's own pipeline glue
(/root/reference/starster/reconstruct.py: `reconstruct_scene` :19-72 and `run_sparse_ga` :75-113) in this container with every
third-party callee replaced by a recorder: which settings does the reference pass down, in which order, and what does it return?
(The optimiser below it, `sparse_scene_optimizer_slam` :116-457, is pinned numerically by tools/gen_align_goldens.py; here it is a
recorder too.)  Only numbers are written: settings as floats / ints / flags, strings as flags of equality.
Run:  python tools/gen_reconstruct_call_goldens.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_align_goldens as gag  # noqa: E402  (its stub set for the absent packages)


def main():
    ref = gag.load_reference()
    log = []
    rec = lambda name, ret=None: (lambda *a, **k: (log.append((name, a, k)), ret(*a, **k) if callable(ret) else ret)[1])
    ref.prepare_images_for_mast3r = rec("prepare_images_for_mast3r", lambda imgs: [dict(idx=i, instance=str(i)) for i in range(len(imgs))])
    ref.make_pairs = rec("make_pairs", lambda imgs, **k: [(a, b) for a in imgs for b in imgs if a is not b])
    ref.convert_dust3r_pairs_naming = rec("convert_dust3r_pairs_naming", lambda imgs, pairs: "PAIRS_IN")
    ref.forward_mast3r = rec("forward_mast3r", lambda pairs_in, model, **k: ("PAIRS", k["cache_path"]))
    ref.prepare_canonical_data = rec("prepare_canonical_data", ("TMP_PAIRS", "SCORES", "CANON_VIEWS", "CANON_PATHS", "PREDS21"))
    ref.compute_min_spanning_tree = rec("compute_min_spanning_tree", "MST")
    ref.condense_data = rec("condense_data", ("IMSIZES", "PPS", "FOCALS", "CORE", "ANCHORS", "CORRES", "CORRES2D", "PREDS21B"))
    ref.sparse_scene_optimizer_slam = rec("sparse_scene_optimizer_slam", ("IMGS", {"coarse": 1}, {"fine": 1}, {"params": 1}))
    ref.SparseGA = rec("SparseGA", lambda *a: ("SPARSEGA",) + a)
    raw = [torch.zeros(3, 16, 16) for _ in range(3)]
    files = ["0.png", "1.png", "2.png"]
    ret = ref.reconstruct_scene("MODEL", raw, files, "cpu", optim_params={"warm": 1}, tmpdir="/tmp/st3r_calls")
    names = [e[0] for e in log]
    order = ["prepare_images_for_mast3r", "make_pairs", "convert_dust3r_pairs_naming", "forward_mast3r", "prepare_canonical_data",
             "compute_min_spanning_tree", "condense_data", "sparse_scene_optimizer_slam", "SparseGA"]
    E = {e[0]: e for e in log}
    out = {"call_order": np.array([order.index(n) for n in names])}
    mp = E["make_pairs"][2]
    out["make_pairs_complete_symmetrize_noprefilter"] = np.array(
        [int(mp.get("scene_graph") == "complete"), int(mp.get("symmetrize") is True), int(mp.get("prefilter") is None)])
    out["n_pairs_of_3_views"] = np.array(len(E["convert_dust3r_pairs_naming"][1][1]))
    out["naming_gets_filelist"] = np.array(int(E["convert_dust3r_pairs_naming"][1][0] == files))
    fm = E["forward_mast3r"]
    out["forward_subsample"] = np.array(fm[2]["subsample"]); out["forward_desc_conf_is_desc_conf"] = np.array(int(fm[2]["desc_conf"] == "desc_conf"))
    out["forward_cache_is_tmpdir"] = np.array(int(fm[2]["cache_path"] == "/tmp/st3r_calls"))
    out["forward_gets_model_and_named_pairs"] = np.array(int(fm[1] == ("PAIRS_IN", "MODEL")))
    pc = E["prepare_canonical_data"]
    out["canon_mode_is_avg_angle"] = np.array(int(pc[2]["mode"] == "avg-angle")); out["canon_subsample"] = np.array(pc[1][2])
    out["canon_gets_filelist_and_pairs"] = np.array(int(pc[1][0] == files and pc[1][1] == "PAIRS"))
    out["mst_gets_scores"] = np.array(int(E["compute_min_spanning_tree"][1] == ("SCORES",)))
    cd = E["condense_data"][1]
    out["condense_args_ok"] = np.array(int(cd[:4] == (files, "TMP_PAIRS", "CANON_VIEWS", "PREDS21") and cd[4] == torch.float32))
    so = E["sparse_scene_optimizer_slam"]
    out["optimizer_positional_ok"] = np.array(int(so[1] == (files, 8, "IMSIZES", "PPS", "FOCALS", "CORE", "ANCHORS", "CORRES", "CORRES2D",
                                                            "PREDS21B", "CANON_PATHS", "MST")))
    kw = so[2]
    out["lr1"], out["niter1"], out["lr2"], out["niter2"] = (np.array(kw[k]) for k in ("lr1", "niter1", "lr2", "niter2"))
    out["opt_depth"] = np.array(int(kw["opt_depth"])); out["matching_conf_thr"] = np.array(kw["matching_conf_thr"])
    out["shared_intrinsics"] = np.array(int(kw["shared_intrinsics"])); out["prev_params_is_optim_params"] = np.array(int(kw["prev_params"] == {"warm": 1}))
    out["optimizer_kw_names_sorted_hash"] = np.array(sorted(kw) == sorted(["lr1", "niter1", "lr2", "niter2", "opt_depth", "matching_conf_thr",
                                                                           "shared_intrinsics", "cache_path", "device", "dtype", "prev_params"]))
    sg = E["SparseGA"][1]
    out["sparsega_gets_fine_result"] = np.array(int(sg == ("IMGS", "PAIRS_IN", {"fine": 1}, "ANCHORS", "CANON_PATHS")))
    out["returns_tuple_scene_params"] = np.array(int(isinstance(ret, tuple) and len(ret) == 2 and ret[1] == {"params": 1} and ret[0][0] == "SPARSEGA"))
    path = os.path.join(ROOT, "tests", "golden", "reconstruct_calls.npz")
    np.savez_compressed(path, **out)
    print("wrote", path); print({k: v.tolist() for k, v in out.items()})


if __name__ == "__main__":
    main()
