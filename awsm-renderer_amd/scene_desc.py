"""
scene_desc.py — the scene description the synthetic-scene generators produce and the host layer consumes.

It plays the role of the reference's `GltfData` after `GltfLoader::into_data` (crates/renderer/src/gltf/data.rs):
decoded accessors per primitive, node TRS tree, skins, materials, decoded RGBA8 images.  No glTF files exist in
this environment (SURVEY.md §7 "Assets"), so scenes are generated in-repo (scenes.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np


@dataclass
class TextureRef:
    texture: int                     # index into SceneDesc.textures
    sampler: int = 0                 # index into SceneDesc.samplers
    uv_index: int = 0
    transform: Optional[dict] = None  # {offset, origin, rotation, scale} -> TextureTransform


@dataclass
class MaterialDesc:
    kind: str = "pbr"                # "pbr" | "unlit"
    base_color_factor: Tuple[float, float, float, float] = (1, 1, 1, 1)
    metallic_factor: float = 1.0
    roughness_factor: float = 1.0
    normal_scale: float = 1.0
    occlusion_strength: float = 1.0
    emissive_factor: Tuple[float, float, float] = (0, 0, 0)
    base_color_tex: Optional[TextureRef] = None
    metallic_roughness_tex: Optional[TextureRef] = None
    normal_tex: Optional[TextureRef] = None
    occlusion_tex: Optional[TextureRef] = None
    emissive_tex: Optional[TextureRef] = None
    double_sided: bool = False
    alpha_mode: str = "opaque"       # "opaque" | "mask" | "blend"  (MaterialAlphaMode, materials.rs:255-273)
    alpha_cutoff: float = 0.5        # Mask only
    debug_bitmask: int = 0
    vertex_color_set: Optional[int] = None
    emissive_strength: Optional[float] = None
    ior: Optional[float] = None
    specular: Optional[dict] = None       # {tex, factor, color_tex, color_factor}
    transmission: Optional[dict] = None   # {tex, factor}; factor > 0 or a texture routes the mesh to the transparent pass
    volume: Optional[dict] = None         # {thickness_tex, thickness_factor, attenuation_distance, attenuation_color}
    clearcoat: Optional[dict] = None      # {tex, factor, roughness_tex, roughness_factor, normal_tex, normal_scale}
    sheen: Optional[dict] = None          # {roughness_tex, roughness_factor, color_tex, color_factor}
    diffuse_transmission: Optional[dict] = None   # {tex, factor, color_tex, color_factor}          (packed, not shaded: pbr.rs:418-447)
    dispersion: Optional[float] = None            #                                                  (pbr.rs:529-532)
    anisotropy: Optional[dict] = None             # {tex, strength, rotation}                        (pbr.rs:534-551)
    iridescence: Optional[dict] = None            # {tex, factor, ior, thickness_tex, thickness_min, thickness_max}   (pbr.rs:553-581)

    def is_transparency_pass(self) -> bool:
        """materials/pbr.rs:213-224, materials/unlit.rs:36-38"""
        if self.alpha_mode in ("blend", "mask"):
            return True
        if self.kind != "unlit" and self.transmission is not None:
            return float(self.transmission.get("factor", 0.0)) > 0.0 or self.transmission.get("tex") is not None
        return False


@dataclass
class PrimitiveDesc:
    positions: np.ndarray            # (V,3) f32
    normals: np.ndarray              # (V,3) f32
    indices: np.ndarray              # (T,3) u32
    material: int = 0
    tangents: Optional[np.ndarray] = None    # (V,4) f32
    uvs: List[np.ndarray] = field(default_factory=list)      # each (V,2) f32
    colors: List[np.ndarray] = field(default_factory=list)   # each (V,4) f32
    joints: List[np.ndarray] = field(default_factory=list)   # per set (V,4) u32
    weights: List[np.ndarray] = field(default_factory=list)  # per set (V,4) f32
    instances: Optional[List[tuple]] = None                 # GPU instancing: [(translation xyz, rotation xyzw, scale xyz), ...] (meshes.rs:176-218)
    morph_targets: List[dict] = field(default_factory=list)  # {positions?, normals?, tangents?} each (V,3) f32
    morph_weights: Optional[np.ndarray] = None               # (targets,) f32  (glTF mesh.weights)
    animated_morph_weights: Optional[np.ndarray] = None      # written through the animation path ([1..n+1))
    hud: bool = False                # Mesh.hud (meshes/mesh.rs:28): drawn by the two HUD passes (render.rs:169-178,301-312) over the world, with a depth buffer of its own;
                                     # carries visibility AND transparency geometry (gltf/buffers/mesh.rs:33-39)


@dataclass
class NodeDesc:
    translation: Tuple[float, float, float] = (0, 0, 0)
    rotation: Tuple[float, float, float, float] = (0, 0, 0, 1)   # xyzw
    scale: Tuple[float, float, float] = (1, 1, 1)
    parent: Optional[int] = None
    primitives: List[PrimitiveDesc] = field(default_factory=list)
    skin: Optional[int] = None


@dataclass
class SkinDesc:
    joints: List[int]                # node indices
    inverse_bind: np.ndarray         # (J,4,4) f32, [j][col][row]


@dataclass
class SceneDesc:
    nodes: List[NodeDesc]
    materials: List[MaterialDesc]
    textures: List[np.ndarray] = field(default_factory=list)   # (h,w,4) u8, already linear-converted where sRGB
    samplers: List[dict] = field(default_factory=list)         # AwsmSampler fields
    skins: List[SkinDesc] = field(default_factory=list)
    lights: List[dict] = field(default_factory=list)           # {kind, color, intensity, direction/position/...}
    width: int = 640
    height: int = 360
    view: np.ndarray = None          # (4,4) f32 [col][row]
    proj: np.ndarray = None
    camera_position: Tuple[float, float, float] = (0, 0, 0)
    skybox_rgba: Tuple[float, float, float, float] = (0, 0, 0, 1)
    prefiltered_rgb: Tuple[float, float, float] = (1, 1, 1)
    irradiance_rgb: Tuple[float, float, float] = (1, 1, 1)
    env_cubes: Optional[dict] = None  # texel cubemaps: {"skybox" | "prefiltered" | "irradiance": [level0, level1, ...]}, each level (6, N_l, N_l, 4) float16,
                                      # faces +X -X +Y -Y +Z -Z; a missing entry keeps the uniform colour above
    prefiltered_mip_count: int = 9   # 256^2 cube with a full chain
    irradiance_mip_count: int = 9
    lut_size: int = 64




# MipmapTextureKind (renderer-core/src/texture/mipmap.rs:28-47) by texture role, as gltf/populate/material.rs assigns them
MIP_KIND_ALBEDO, MIP_KIND_NORMAL, MIP_KIND_METALLIC_ROUGHNESS, MIP_KIND_OCCLUSION, MIP_KIND_EMISSIVE = 0, 1, 2, 3, 4
MIP_KIND_SPECULAR, MIP_KIND_SPECULAR_COLOR, MIP_KIND_TRANSMISSION, MIP_KIND_VOLUME_THICKNESS = 5, 6, 7, 8


def texture_mip_kinds(scene: "SceneDesc") -> List[int]:
    """Mip-generation filter per texture of the scene: the role under which the image first enters the pool
    (materials in order; roles in the order pbr_material_mapper visits them, populate/material.rs:94-640)."""
    kinds: List[Optional[int]] = [None] * len(scene.textures)

    def use(ref, kind):
        if ref is not None and 0 <= ref.texture < len(kinds) and kinds[ref.texture] is None:
            kinds[ref.texture] = kind

    for m in scene.materials:
        use(m.base_color_tex, MIP_KIND_ALBEDO)
        use(m.metallic_roughness_tex, MIP_KIND_METALLIC_ROUGHNESS)
        use(m.normal_tex, MIP_KIND_NORMAL)
        use(m.occlusion_tex, MIP_KIND_OCCLUSION)
        use(m.emissive_tex, MIP_KIND_EMISSIVE)
        if m.specular:
            use(m.specular.get("tex"), MIP_KIND_SPECULAR)
            use(m.specular.get("color_tex"), MIP_KIND_SPECULAR)
        if m.transmission:
            use(m.transmission.get("tex"), MIP_KIND_TRANSMISSION)
        if m.volume:
            use(m.volume.get("thickness_tex"), MIP_KIND_VOLUME_THICKNESS)
        if m.clearcoat:
            use(m.clearcoat.get("tex"), MIP_KIND_ALBEDO)
            use(m.clearcoat.get("roughness_tex"), MIP_KIND_METALLIC_ROUGHNESS)
            use(m.clearcoat.get("normal_tex"), MIP_KIND_NORMAL)
        if m.sheen:
            use(m.sheen.get("color_tex"), MIP_KIND_SPECULAR)
            use(m.sheen.get("roughness_tex"), MIP_KIND_METALLIC_ROUGHNESS)
    return [MIP_KIND_ALBEDO if k is None else k for k in kinds]
