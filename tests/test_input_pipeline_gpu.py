"""GPU: ToTensor + Normalize + pad + mask on the device (csrc/preprocess.hip) against the
reference's host-side sequence (/root/reference/datasets/da_transforms.py:250-276,
/root/reference/util/misc.py:387-409) -- bit-exact."""
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
