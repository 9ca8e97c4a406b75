"""Random test scenes shared by the CPU and GPU suites: triangle soups with every material class the path
distinguishes (opaque, MASK, BLEND stochastic, thick / thin transmission, textured PBR), plus local lights."""
import math

import numpy as np

from hobbyrenderer_amd import scenes, structs as S


def _texture(rng, size, kind):
    t = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
    if kind == "normal":
        t[..., 0:2] = rng.integers(96, 160, (size, size, 2), dtype=np.uint8)
    if kind == "alpha":
        t[..., 3] = (rng.random((size, size)) > 0.5) * 255
    return t


def random_soup(luts, n_tris, seed, blend_fraction=0.0, mask_fraction=0.0, textured=False, lights=True):
    rng = np.random.default_rng(seed)
    b = scenes.SceneBuilder()
    n_mesh = 6
    per = max(1, n_tris // n_mesh)
    mats = [b.add_material(m_BaseColor=tuple(rng.uniform(0.2, 0.9, 3)) + (1.0,))]
    if textured:
        ta, tn, tr, te, tm = (b.add_texture(_texture(rng, 16, k)) for k in ("albedo", "normal", "orm", "emissive", "alpha"))
        mats.append(b.add_material(m_TextureFlags=S.TEXFLAG_ALBEDO | S.TEXFLAG_NORMAL | S.TEXFLAG_ROUGHNESS_METALLIC | S.TEXFLAG_EMISSIVE,
                                   m_AlbedoTextureIndex=ta, m_NormalTextureIndex=tn, m_RoughnessMetallicTextureIndex=tr, m_EmissiveTextureIndex=te,
                                   m_EmissiveFactor=(0.5, 0.4, 0.3, 1), m_AlbedoSamplerIndex=1, m_NormalSamplerIndex=0, m_RoughnessSamplerIndex=1,
                                   m_EmissiveSamplerIndex=0))
        mats.append(b.add_material(m_TextureFlags=S.TEXFLAG_ALBEDO, m_AlbedoTextureIndex=tm, m_AlphaMode=S.ALPHA_MODE_MASK, m_AlphaCutoff=0.5,
                                   m_AlbedoSamplerIndex=1))
        mats.append(b.add_material(m_TextureFlags=S.TEXFLAG_ALBEDO, m_AlbedoTextureIndex=ta, m_AlphaMode=S.ALPHA_MODE_BLEND, m_AlbedoSamplerIndex=0,
                                   m_RoughnessMetallic=(0.3, 1.0)))
    if blend_fraction:
        mats.append(b.add_material(m_AlphaMode=S.ALPHA_MODE_BLEND, m_BaseColor=(0.8, 0.6, 0.4, 0.5)))
        mats.append(b.add_material(m_AlphaMode=S.ALPHA_MODE_BLEND, m_TransmissionFactor=1.0, m_IOR=1.5, m_SigmaA=(0.5, 0.2, 0.1),
                                   m_RoughnessMetallic=(0.05, 0.0)))
        mats.append(b.add_material(m_AlphaMode=S.ALPHA_MODE_BLEND, m_TransmissionFactor=0.7, m_IsThinSurface=1, m_RoughnessMetallic=(0.4, 0.0)))
        mats.append(b.add_material(m_AlphaMode=S.ALPHA_MODE_OPAQUE, m_TransmissionFactor=0.9, m_IOR=1.33, m_RoughnessMetallic=(0.2, 0.0),
                                   m_SigmaA=(0.1, 0.3, 0.6)))
    if mask_fraction:
        mats.append(b.add_material(m_AlphaMode=S.ALPHA_MODE_MASK, m_BaseColor=(0.5, 0.9, 0.5, 0.3), m_AlphaCutoff=0.5))
        mats.append(b.add_material(m_AlphaMode=S.ALPHA_MODE_MASK, m_BaseColor=(0.5, 0.9, 0.5, 0.9), m_AlphaCutoff=0.5))
    for m in range(n_mesh):
        verts, idx = [], []
        for t in range(per):
            c = rng.uniform(-1.5, 1.5, 3)
            p = c + rng.uniform(-0.5, 0.5, (3, 3))
            nrm = np.cross(p[1] - p[0], p[2] - p[0]); nrm /= np.linalg.norm(nrm) + 1e-12
            tan = p[1] - p[0]; tan /= np.linalg.norm(tan) + 1e-12
            for k in range(3):
                verts.append(scenes.quantize_vertex(p[k], nrm, rng.uniform(-0.5, 2.5, 2), tan, 1.0 if (t & 1) else -1.0))
            idx += [3 * t, 3 * t + 1, 3 * t + 2]
        mesh = b.add_mesh(np.array(verts, S.VertexQuantized), np.array(idx, np.uint32))
        mat = mats[m % len(mats)]
        ang = rng.uniform(0, 2 * math.pi)
        rot = [[math.cos(ang), 0, -math.sin(ang)], [0, 1, 0], [math.sin(ang), 0, math.cos(ang)]]
        b.add_instance(mesh, mat, scenes._mat(tuple(rng.uniform(0.5, 1.5, 3)), rot, tuple(rng.uniform(-0.5, 0.5, 3))))
    if lights:
        b.add_light(S.LIGHT_POINT, position=(0.3, 2.5, -0.4), color=(1, 0.9, 0.8), intensity=20.0, radius=0.1)
    return b.finalize(luts)


# Material classes the path distinguishes; a random SUBSET per scene makes the upload-time scene traits (what the kernels specialise
# on: medium tracking, stochastic alpha, textures, non-opaque geometry, local lights) vary independently of each other.
_MATERIAL_CLASSES = [
    dict(),                                                                                                   # opaque default
    dict(m_RoughnessMetallic=(0.25, 1.0), m_BaseColor=(0.9, 0.7, 0.3, 1.0)),                                  # opaque metal
    dict(m_EmissiveFactor=(3.0, 2.0, 1.0, 1.0)),                                                              # emissive
    dict(m_AlphaMode=S.ALPHA_MODE_MASK, m_BaseColor=(0.5, 0.9, 0.5, 0.3), m_AlphaCutoff=0.5),                 # MASK, rejected
    dict(m_AlphaMode=S.ALPHA_MODE_MASK, m_BaseColor=(0.5, 0.9, 0.5, 0.9), m_AlphaCutoff=0.5),                 # MASK, accepted
    dict(m_AlphaMode=S.ALPHA_MODE_BLEND, m_BaseColor=(0.8, 0.6, 0.4, 0.5)),                                   # stochastic BLEND, thick
    dict(m_AlphaMode=S.ALPHA_MODE_BLEND, m_BaseColor=(0.8, 0.6, 0.4, 0.4), m_IsThinSurface=1),                # stochastic BLEND, thin
    dict(m_AlphaMode=S.ALPHA_MODE_BLEND, m_TransmissionFactor=1.0, m_IOR=1.5, m_SigmaA=(0.5, 0.2, 0.1), m_RoughnessMetallic=(0.05, 0.0)),   # glass, thick
    dict(m_AlphaMode=S.ALPHA_MODE_BLEND, m_TransmissionFactor=0.7, m_IsThinSurface=1, m_RoughnessMetallic=(0.4, 0.0)),                     # rough thin glass
    dict(m_AlphaMode=S.ALPHA_MODE_OPAQUE, m_TransmissionFactor=0.9, m_IOR=1.33, m_RoughnessMetallic=(0.2, 0.0), m_SigmaA=(0.1, 0.3, 0.6)),  # opaque-flagged transmissive, thick
    dict(m_AlphaMode=S.ALPHA_MODE_OPAQUE, m_TransmissionFactor=0.6, m_IsThinSurface=1),                       # opaque-flagged transmissive, thin
]


def random_trait_scene(luts, seed, n_tris=160):
    """A triangle soup over a random subset of the material classes, with a random light set (sun only / + point / + spot / + both)."""
    rng = np.random.default_rng(1000 + seed)
    b = scenes.SceneBuilder()
    classes = rng.choice(len(_MATERIAL_CLASSES), size=int(rng.integers(1, 4)), replace=False)
    mats = [b.add_material(**_MATERIAL_CLASSES[int(k)]) for k in classes]
    n_mesh = 5
    per = max(1, n_tris // n_mesh)
    for m in range(n_mesh):
        verts, idx = [], []
        for t in range(per):
            c = rng.uniform(-1.5, 1.5, 3)
            p = c + rng.uniform(-0.6, 0.6, (3, 3))
            nrm = np.cross(p[1] - p[0], p[2] - p[0]); nrm /= np.linalg.norm(nrm) + 1e-12
            tan = p[1] - p[0]; tan /= np.linalg.norm(tan) + 1e-12
            for k in range(3):
                verts.append(scenes.quantize_vertex(p[k], nrm, rng.uniform(0.0, 1.0, 2), tan, 1.0))
            idx += [3 * t, 3 * t + 1, 3 * t + 2]
        mesh = b.add_mesh(np.array(verts, S.VertexQuantized), np.array(idx, np.uint32))
        b.add_instance(mesh, mats[m % len(mats)], scenes._mat(tuple(rng.uniform(0.6, 1.4, 3)), None, tuple(rng.uniform(-0.4, 0.4, 3))))
    lights = int(rng.integers(0, 4))
    if lights & 1:
        b.add_light(S.LIGHT_POINT, position=(0.3, 2.5, -0.4), color=(1, 0.9, 0.8), intensity=20.0, radius=0.1)
    if lights & 2:
        b.add_light(S.LIGHT_SPOT, position=(-1.0, 2.0, -2.0), direction=(0.35, -0.7, 0.6), intensity=30.0, range_=20.0, radius=0.05, inner=0.3, outer=0.6)
    return b.finalize(luts), [int(k) for k in classes], lights
