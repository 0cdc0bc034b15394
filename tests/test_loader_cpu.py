"""Decode-ahead loader (vpr_amd/loader.py) on the CPU device: same bytes, batches and order as the serial PIL loop."""
import os

import numpy as np
import pytest
import torch
from PIL import Image


def _write_images(tmp_path, n, sizes=((64, 48), (32, 32))):
    rng = np.random.default_rng(0)
    names = []
    for i in range(n):
        W, H = sizes[(i // 3) % len(sizes)]                       # runs of 3 per size: groups interleave in the file list
        arr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        name = f"im_{i:03d}.{'png' if i % 2 else 'jpg'}"
        img = Image.fromarray(arr)
        if i % 7 == 3:
            img, name = img.convert("L"), f"im_{i:03d}.png"            # grey, RGBA and palette files go through convert("RGB")
        elif i % 11 == 5:
            img, name = img.convert("RGBA"), f"im_{i:03d}.png"
        elif i % 13 == 6:
            img, name = img.convert("P"), f"im_{i:03d}.png"
        img.save(os.path.join(tmp_path, name), quality=90)
        names.append(name)
    return names


def _serial(image_dir, names):
    return {f: np.asarray(Image.open(os.path.join(image_dir, f)).convert("RGB")) for f in names}


@pytest.mark.parametrize("workers,depth,export", [(1, 1, "raw"), (4, 3, "raw"), (8, 2, "auto"), (4, 2, "arrow")])
def test_loader_equals_serial_decode(tmp_path, workers, depth, export):
    from vpr_amd.loader import ImageBatchLoader, _arrow_export_works, batches_by_size
    if export == "arrow" and not _arrow_export_works():
        pytest.skip("pyarrow / Pillow Arrow export not available here")
    names = _write_images(str(tmp_path), 23)
    ref = _serial(str(tmp_path), names)
    plan = batches_by_size(str(tmp_path), names, 4)
    assert [s for s, _, _ in plan][:2] == [(64, 48)] * 2           # first size first, in order of first appearance
    seen = []
    ld = ImageBatchLoader(str(tmp_path), names, 4, "cpu", workers=workers, depth=depth, export=export)
    assert len(ld) == len(plan) and ld.export == ("arrow" if export == "arrow" else "raw")      # auto on a CPU device: raw
    for (size, idxs_p, names_p), (idxs, bnames, u8) in zip(plan, ld):
        assert idxs == idxs_p and bnames == names_p
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (len(idxs), size[1], size[0], 3)
        for j, f in enumerate(bnames):
            assert np.array_equal(u8[j].numpy(), ref[f]), f
            assert names[idxs[j]] == f
        seen += idxs
    assert sorted(seen) == list(range(len(names)))
    # a second pass over the same loader object decodes again (fresh ring, same result)
    assert sum(len(i) for i, _, _ in ld) == len(names)


def test_loader_reports_a_broken_file(tmp_path):
    from vpr_amd.loader import ImageBatchLoader
    names = _write_images(str(tmp_path), 6, sizes=((16, 16),))
    with open(os.path.join(tmp_path, names[4]), "r+b") as f:        # keep the header (size probe passes), cut the data
        f.truncate(60)
    with pytest.raises(Exception):
        for _ in ImageBatchLoader(str(tmp_path), names, 2, "cpu", workers=2):
            pass


def test_loader_arrow_export_falls_back_for_multi_block_images(tmp_path):
    """Images above PIL's 16 MB block size are stored in several blocks and have no zero-copy Arrow view: the Arrow
    export path falls back to the raw export for those files (same bytes)."""
    from vpr_amd.loader import ImageBatchLoader, _arrow_export_works
    if not _arrow_export_works():
        pytest.skip("pyarrow / Pillow Arrow export not available here")
    rng = np.random.default_rng(2)
    big = rng.integers(0, 256, (2400, 2000, 3), dtype=np.uint8)          # 19.2 MB as RGBX: two blocks
    Image.fromarray(big).save(os.path.join(tmp_path, "big0.png"))
    Image.fromarray(big[::-1].copy()).save(os.path.join(tmp_path, "big1.png"))
    out = list(ImageBatchLoader(str(tmp_path), ["big0.png", "big1.png"], 2, "cpu", workers=2, export="arrow"))
    assert len(out) == 1 and np.array_equal(out[0][2][0].numpy(), big) and np.array_equal(out[0][2][1].numpy(), big[::-1])


def test_truncated_jpeg_is_skipped_by_the_test_split_filter(tmp_path, capsys):
    """evaluate._loadable(decode=True): a JPEG cut in the middle of its scan data passes PIL's verify() but fails at
    decode; the reference's TestImageDataset.__getitem__ (val_and_test_swin_2.py:150-165) returns None for it and the
    collate drops the item — the test-split filter must skip it too instead of letting the loader abort the run."""
    import numpy as np
    from PIL import Image
    from vpr_amd import evaluate
    good, bad = tmp_path / "good.jpg", tmp_path / "bad.jpg"
    Image.fromarray(np.random.default_rng(0).integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(good, quality=90)
    data = good.read_bytes()
    bad.write_bytes(data[: len(data) // 2])
    assert evaluate._loadable(str(good), decode=True)
    assert evaluate._loadable(str(bad)) is True                      # verify() alone does not notice
    assert evaluate._loadable(str(bad), decode=True) is False
    assert "Skipping invalid/corrupt image file" in capsys.readouterr().out
