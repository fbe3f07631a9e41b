"""GPU: ToTensor + Normalize + pad + mask on the device (csrc/preprocess.hip) against the
reference's host-side sequence (/root/reference/datasets/da_transforms.py:250-276,
/root/reference/util/misc.py:387-409) -- bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels_last", [True, False])
@pytest.mark.parametrize("sizes", [[(37, 53), (37, 53)], [(64, 48), (30, 80), (1, 1)]])
def test_collate_uint8_matches_reference_sequence(sizes, channels_last):
    from datr_amd.input_pipeline import IMAGENET_MEAN, IMAGENET_STD, collate_uint8_on_device
    from datr_amd.nested import nested_tensor_from_tensor_list
    g = torch.Generator().manual_seed(len(sizes))
    imgs = [torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8) for h, w in sizes]
    out = collate_uint8_on_device(imgs, device="cuda:0", channels_last=channels_last)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    ref = nested_tensor_from_tensor_list(
        [(im.permute(2, 0, 1).float().div(255) - mean) / std for im in imgs])     # to_tensor, normalize
    assert torch.equal(out.tensors.cpu(), ref.tensors)
    assert torch.equal(out.mask.cpu(), ref.mask)
    assert out.padded == ref.padded
    assert out.tensors.is_contiguous(memory_format=torch.channels_last if channels_last
                                     else torch.contiguous_format)


@pytest.mark.parametrize("hw,target,flip", [((375, 500), (480, 640), False), ((1024, 2048), (800, 1600), True),
                                            ((480, 640), (480, 640), True), ((600, 901), (400, 901), False),
                                            ((333, 517), (800, 1242), True), ((97, 131), (31, 45), False),
                                            ((64, 64), (64, 200), True)])
def test_device_resize_and_flip_match_pillow_bit_for_bit(hw, target, flip):
    """csrc/resize.hip against Pillow itself (the reference's F.hflip + F.resize on PIL images,
    da_transforms.py:62-140): identical uint8 pixels for up- and down-scaling, size-preserving
    axes (Pillow skips that pass), with and without the flip."""
    from PIL import Image
    from datr_amd.input_pipeline import resize_uint8_on_device
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    img = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    img[: hw[0] // 3] = (img[: hw[0] // 3] // 64) * 64 + 31            # flat patches: rounding boundaries
    pil = Image.fromarray(img)
    if flip:
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
    want = np.asarray(pil.resize((target[1], target[0]), Image.BILINEAR))
    got = resize_uint8_on_device(torch.from_numpy(img).to("cuda:0"), target, flip=flip).cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.uint8
    assert np.array_equal(got, want), f"{(got != want).sum()} pixels differ, max {np.abs(got.astype(int) - want).max()}"
