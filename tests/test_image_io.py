"""PNG writer of the generation harness (reference `src/misc/image_io.py:42-73`)."""
import numpy as np
import torch

from mv_ldm_amd.image_io import decode_png, encode_png, prep_image, save_image


def test_prep_image_truncates_like_the_reference():
    x = torch.tensor([[[0.0, 0.5, 0.999, 1.0, 1.7, -0.3]]]).expand(3, 1, 6)
    p = prep_image(x)
    assert p.dtype == np.uint8 and p.shape == (1, 6, 3)
    assert p[0, :, 0].tolist() == [0, 127, 254, 255, 255, 0]            # (x.clip(0,1) * 255).type(uint8): truncation
    assert prep_image(torch.rand(5, 7)).shape == (5, 7, 3) and prep_image(torch.rand(4, 5, 7)).shape == (5, 7, 4)


def test_png_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    for shape in ((3, 17, 23), (4, 8, 8), (1, 5, 9)):
        img = torch.rand(shape, generator=g)
        path = tmp_path / "a" / "b" / f"{shape[0]}.png"      # parent directories are created
        save_image(img, path)
        data = path.read_bytes()
        assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[-8:-4] == b"IEND"
        assert np.array_equal(decode_png(data), prep_image(img))
    px = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    assert np.array_equal(decode_png(encode_png(px)), px)
