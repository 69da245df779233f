"""Scene factories with the constants of the reference's scene/scene_registry.py:18-79,
plus the two benchmark scenes of SURVEY.md section 8(d) (configs 4 and 5)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .primitives import SDFBox, SDFLine, SDFSphere, SDFTorus
from .transformations import SDFAffineTransformation, SDFOnion, SDFSmoothUnion, SDFUnion


def make_room():
    """The closing shell used by make_test_scene2: Onion(Box 5^3, 0.1)."""
    return SDFOnion(SDFBox(halfsides=(5.0, 5.0, 5.0)), radius=0.1)


def make_test_scene():
    """Smooth union (k=22) of an affine onion-box, an affine sphere, a capsule and an
    affine torus: 4 leaves, 3 affine nodes, open scene."""
    return SDFSmoothUnion(
        sdfs=[
            SDFAffineTransformation(SDFOnion(SDFBox(halfsides=(0.1, 0.2, 0.05)), radius=0.1),
                                    orientation=[0.9014, 0.25, 0.25, 0.25], translation=[0.0, 0.25, 0.25]),
            SDFAffineTransformation(SDFSphere(radius=0.5),
                                    orientation=[1.0, 0.0, 0.0, 0.0], translation=[0.0, 0.0, 1.0]),
            SDFLine(start=(-1.0, 1.0, 2.0), end=(1.0, 1.0, 0.0), radius=0.1),
            SDFAffineTransformation(SDFTorus(radius1=0.5, radius2=0.1),
                                    orientation=[0.0, 0.5 ** 0.5, 0.5 ** 0.5, 0.0], translation=[0.0, 0.5, 1.0]),
        ],
        blend_k=22.0,
    )


def make_test_scene2():
    """The default scene of the reference's main.py: room shell + union(sphere, torus, capsule)."""
    return SDFUnion([
        make_room(),
        SDFUnion(sdfs=[
            SDFSphere(radius=0.5),
            SDFTorus(radius1=1.0, radius2=0.25),
            SDFLine(start=(1.0, 0.0, 0.0), end=(-1.0, 0.0, 0.0), radius=0.1),
        ]),
    ])


def make_closed_test_scene():
    """Config 4: make_test_scene() closed by the room so every ray hits."""
    return SDFUnion([make_test_scene(), make_room()])


def make_many_primitive_scene(n_prims: int = 32, seed: int = 1234):
    """Config 5: room + smooth union of ``n_prims`` affine-wrapped primitives cycling
    sphere / box / torus / capsule; poses and sizes from torch seed ``seed``."""
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(n_prims, 3, generator=g) * 6.0 - 3.0
    q = F.normalize(torch.randn(n_prims, 4, generator=g), dim=-1)
    s = torch.rand(n_prims, 4, generator=g) * 0.4 + 0.1
    kids = []
    for i in range(n_prims):
        a, b, c, d = (float(x) for x in s[i])
        which = i % 4
        if which == 0:
            prim = SDFSphere(a)
        elif which == 1:
            prim = SDFBox((a, b, c))
        elif which == 2:
            prim = SDFTorus(a + 0.2, 0.3 * b)
        else:
            prim = SDFLine((-a, 0.0, 0.0), (b, c, 0.0), 0.5 * d)
        kids.append(SDFAffineTransformation(prim, orientation=q[i].tolist(), translation=t[i].tolist()))
    return SDFUnion([make_room(), SDFSmoothUnion(kids, blend_k=22.0)])
