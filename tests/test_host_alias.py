"""`import starster` is literally drop-in: the names of the reference's package (starster/__init__.py:1-9,
docs/api.rst:1-37) resolve to the MI355X implementation, submodules included.  CPU only."""
import inspect


def test_starster_alias_exports_reference_names():
    import starst3r_amd
    import starster
    for name in ("Scene", "reconstruct_scene", "load_image", "load_images", "process_image", "interp_se3",
                 "interp_se3_path", "Mast3rModel", "__version__"):
        assert hasattr(starster, name), name
        assert getattr(starster, name) is getattr(starst3r_amd, name), name
    for fn in ("init_3dgs", "render_3dgs", "render_3dgs_original", "run_3dgs_optim"):
        assert getattr(starster.gs, fn) is getattr(starst3r_amd.gs, fn)
    assert starster.gs.train is starst3r_amd.gs.run_3dgs_optim        # north_star wording


def test_starster_submodules_import_like_the_reference():
    import starster.gs, starster.image, starster.reconstruct, starster.scene, starster.utils  # noqa: E401,F401
    from starster.gs import run_3dgs_optim  # noqa: F401
    from starster.scene import Scene
    from starster.reconstruct import reconstruct_scene, sparse_scene_optimizer_slam  # noqa: F401
    import starster
    assert Scene is starster.Scene
    assert inspect.ismodule(starster.reconstruct) and callable(starster.reconstruct)
    # same public signature as the reference (starster/reconstruct.py:21)
    assert list(inspect.signature(reconstruct_scene).parameters) == ["model", "imgs", "filelist", "device",
                                                                     "optim_params", "tmpdir"]
