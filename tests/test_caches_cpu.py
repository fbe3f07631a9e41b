"""Geometry caches of datr_amd.msda must never serve another pyramid's entry:
they are keyed by the geometry tensor's storage address, which is an identity only while the
tensor lives -- the entries therefore hold the tensor.  The scenario of the finding: many
different geometries created and dropped (multi-scale training, teacher + student with their own
transformer-level caches), the allocator recycling addresses."""
import random

import numpy as np
import torch


def _geometry(rng):
    h, w = rng.randint(8, 200), rng.randint(8, 200)
    shapes = [(h, w), ((h + 1) // 2, (w + 1) // 2), ((h + 3) // 4, (w + 3) // 4), ((h + 7) // 8, (w + 7) // 8)]
    return shapes


def test_inverse_wh_and_host_meta_survive_address_recycling():
    from datr_amd import msda
    msda._INV_WH.clear()
    msda._HOST_META.clear()
    rng = random.Random(0)
    seen_ptrs = set()
    recycled = 0
    for it in range(600):
        shapes = _geometry(rng)
        t = torch.as_tensor(shapes, dtype=torch.long)          # fresh tensor, dropped after the loop body
        lsi = torch.cat((t.new_zeros((1,)), t.prod(1).cumsum(0)[:-1]))
        recycled += t.data_ptr() in seen_ptrs
        seen_ptrs.add(t.data_ptr())
        inv = msda._inverse_wh(t, 8, 4).view(8, 4, 4, 2)
        want = torch.tensor([[1.0 / w, 1.0 / h] for h, w in shapes])
        torch.testing.assert_close(inv[3, :, 2, :], want, rtol=0, atol=0)
        sh_host, ls_host = msda._host_meta(t, lsi)
        assert np.array_equal(sh_host, np.asarray(shapes)) and np.array_equal(ls_host, lsi.numpy())
        del t, lsi
    # the caches are bounded
    assert len(msda._INV_WH) <= 65 and len(msda._HOST_META) <= 257


def test_transformer_level_meta_is_keyed_by_value():
    from datr_amd.transformer import DeformableTransformer
    tr = DeformableTransformer.__new__(DeformableTransformer)
    tr._meta_cache = {}
    a = tr._level_meta([(10, 12), (5, 6)], torch.device("cpu"))
    b = tr._level_meta([(10, 12), (5, 6)], torch.device("cpu"))
    c = tr._level_meta([(10, 13), (5, 7)], torch.device("cpu"))
    assert a[0] is b[0] and a[0] is not c[0]
    assert c[1].tolist() == [0, 130]
