#!/usr/bin/env python
"""Generates tests/golden/scene_add_images.npz by RUNNING the reference's own `Scene`
(/root/reference/starster/scene.py:19-183: constructor, `add_images` :97-155, the flat / inverse properties :79-95) in this
container.  The file is executed from where it lies; its two relative imports are satisfied by stub modules registered as
`starster.gs` (the four names it star-imports, as recorders) and `starster.reconstruct` (`reconstruct_scene` =
tests/fake_reconstruct.Recorder, a deterministic stand-in for Mast3r + alignment).  Pinned by execution: the fake file
names, the warm-start hand-over between calls, that ALL images are re-solved and poses / points replaced, the
`conf > conf_thres` mask, colours from the scaled images, `clean_depth=True`.  Only numeric arrays are written.
Run:  python tools/gen_scene_goldens.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/starster/scene.py"
import fake_reconstruct as fr  # noqa: E402


def load_reference_scene(rec, gs_log):
    pkg = types.ModuleType("starster"); pkg.__path__ = []
    gs = types.ModuleType("starster.gs")
    gs.__all__ = ("init_3dgs", "render_3dgs", "render_3dgs_original", "run_3dgs_optim")
    for name in gs.__all__:
        setattr(gs, name, (lambda n: (lambda *a, **k: gs_log.append((n, len(a), tuple(sorted(k)))) or n))(name))
    recon = types.ModuleType("starster.reconstruct")
    recon.reconstruct_scene = rec
    sys.modules.update({"starster": pkg, "starster.gs": gs, "starster.reconstruct": recon})
    spec = importlib.util.spec_from_file_location("starster.scene", REF)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "starster"
    sys.modules["starster.scene"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    rec, gs_log = fr.Recorder(), []
    ref = load_reference_scene(rec, gs_log)
    out = {}
    sc = ref.Scene(cache_dir="/tmp/st3r_scene_golden", device="cpu")
    out["init_none"] = np.array([int(getattr(sc, k) is None) for k in
                                 ("c2w", "intrinsics", "optim_params", "gs_params", "gs_optims", "gs_strategy", "gs_state")])
    for tag, (k0, k1, kw) in (("a", (0, 2, {})), ("b", (2, 3, dict(conf_thres=2.0)))):
        sc.add_images("MODEL", fr.raw_images(k0, k1), **kw)
        n = len(sc.imgs)
        out[f"{tag}_n_raw"] = np.array(len(sc.raw_imgs)); out[f"{tag}_n_imgs"] = np.array(n)
        out[f"{tag}_imgs"] = np.stack([np.asarray(im) for im in sc.imgs])
        out[f"{tag}_c2w"] = sc.c2w.numpy().copy(); out[f"{tag}_intrinsics"] = sc.intrinsics.numpy().copy()
        out[f"{tag}_w2c"] = sc.w2c.numpy().copy()
        out[f"{tag}_optim_params_call"] = np.array(sc.optim_params["call"])
        out[f"{tag}_pts_counts"] = np.array([p.shape[0] for p in sc.dense_pts])
        out[f"{tag}_pts_flat"] = sc.dense_pts_flat.numpy().copy()
        out[f"{tag}_cols_flat"] = sc.dense_cols_flat.numpy().copy()
    c = rec.calls
    out["call_n_imgs"] = np.array([x["n_imgs"] for x in c])
    out["call_filelist_is_index_png"] = np.array([int(x["filelist"] == [f"{i}.png" for i in range(x["n_imgs"])]) for x in c])
    out["call_optim_params_in"] = np.array([-1 if x["optim_params_in"] is None else x["optim_params_in"] for x in c])
    out["call_tmpdir_is_cache_dir"] = np.array([int(x["tmpdir"] == "/tmp/st3r_scene_golden") for x in c])
    out["call_device_is_cpu"] = np.array([int(x["device"] == "cpu") for x in c])
    out["call_model_passed"] = np.array([int(x["model"] == "MODEL") for x in c])
    out["dense_clean_depth"] = np.array([int(r.dense_calls == [(True, ())]) for r in rec.results])
    # the four 3DGS methods are pass-throughs (scene.py:157-183): positional argument counts the module functions receive
    sc.init_3dgs(); sc.init_3dgs(1e-2, 2e-3); sc.render_3dgs(1, 2, 3, 4); sc.render_3dgs_original(5, 6)
    sc.run_3dgs_optim(7); sc.run_3dgs_optim(7, True, 0.3, 0.02, 0.03, True)
    out["gs_passthrough_argc"] = np.array([e[1] for e in gs_log])
    out["gs_passthrough_name"] = np.array([("init_3dgs", "render_3dgs", "render_3dgs_original", "run_3dgs_optim").index(e[0])
                                           for e in gs_log])
    path = os.path.join(ROOT, "tests", "golden", "scene_add_images.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("pts_counts")}, out["call_optim_params_in"],
          out["gs_passthrough_argc"])


if __name__ == "__main__":
    main()
