"""CPU test of the dataset-preparation counterpart (gan-leaks_amd/z_split.py) against a run of the reference's own z_split.main on
the same miniature CelebA (tests/golden/zsplit_case.npz, made by tests/golden/make_golden.py --zsplit-only with numpy's global
seed 123): identical file names in the three output directories and identical pixels in every file, random crops included."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden_tools():
    spec = importlib.util.spec_from_file_location("make_golden_tools", os.path.join(HERE, "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_split_matches_reference(tmp_path, golden_dir):
    import ganleaks_amd  # noqa: F401
    from ganleaks_amd import z_split
    tools = _golden_tools()
    g = np.load(os.path.join(golden_dir, "zsplit_case.npz"))
    args = tools.zsplit_case(str(tmp_path))
    os.makedirs(args.output_dir1)
    open(os.path.join(args.output_dir1, "stale.png"), "w").close()            # output directories are wiped first, as in the reference
    np.random.seed(123)
    private, public = z_split.main(args)
    assert len(private) == 10 and len(public) == 10 and not set(private) & set(public)
    for key, d in (("train", args.output_dir0), ("pos", args.output_dir1), ("neg", args.output_dir2)):
        names, crcs = tools.dir_digest(d)
        assert names == list(g[key + "_names"])
        assert np.array_equal(np.array(crcs, np.uint32), g[key + "_crc"])
    assert len(os.listdir(args.output_dir0)) == 30
    a = z_split.parse_arguments([])
    assert (a.num_images, a.num_same_id, a.img_size, a.output_dir1) == (10020, 30, 64, "data/celebAhuge_positive")
    args.num_images = 31
    with pytest.raises(AssertionError):
        z_split.main(args)
