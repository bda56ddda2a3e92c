#!/usr/bin/env python3
"""Generates the four 8-bit PNGs of tests/golden/scenes/material-maps/ (data only, synthetic):

  albedo.png    smooth colour ramp + stripes         -> Kd of an uber material (normalized image => gamma flag, TracerBoy.cpp:205-209)
  normal.png    tangent-space normal map (bumps)     -> Material.normalMapIndex   (GetDetailNormal, RayGenCommon.h:273-295)
  specular.png  g = roughness, b = metallic mask     -> Material.specularMapIndex (RayGenCommon.h:331-339)
  emissive.png  glowing dots on black                -> Material.emissiveIndex    (RayGenCommon.h:325-329)

  python tests/golden/make_material_maps.py
"""
import os

import numpy as np

from make_image_fixtures import write_png

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes", "material-maps")


def main():
    os.makedirs(OUT, exist_ok=True)
    n = 32
    y, x = np.mgrid[0:n, 0:n].astype(np.float64)
    u, v = (x + 0.5) / n, (y + 0.5) / n
    albedo = np.stack([0.15 + 0.8 * u, 0.2 + 0.7 * v, 0.9 - 0.6 * u * v], -1)
    albedo[(x.astype(int) // 4) % 2 == 0] *= 0.55
    nx = 0.35 * np.sin(2 * np.pi * 3 * u) * np.cos(2 * np.pi * 2 * v)
    ny = 0.35 * np.cos(2 * np.pi * 2 * u) * np.sin(2 * np.pi * 3 * v)
    normal = np.stack([0.5 - nx / 2, 0.5 - ny / 2, np.full_like(u, 1.0)], -1)   # the shader decodes (0.5 - c) * 2
    normal[:4, :4, :2] = [0.0, 1.0]                                          # |xy| > 1: exercises the max(z, 0.02) clamp (sqrt of a negative number)
    specular = np.stack([np.zeros_like(u), 0.03 + 0.6 * v, (u > 0.5).astype(np.float64)], -1)   # left half dielectric, right half metallic; top rows below the 0.05 "perfect" roughness
    r2 = ((u * 4) % 1 - 0.5) ** 2 + ((v * 4) % 1 - 0.5) ** 2
    emissive = np.where(r2[..., None] < 0.05, np.stack([3.0 * u, 2.0 * v, 1.5 * (1 - u)], -1), 0.0)
    emissive = emissive / 3.0                                                    # stored normalized; dots up to 1.0
    for name, img in (("albedo", albedo), ("normal", normal), ("specular", specular), ("emissive", emissive)):
        s = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint16)
        write_png(os.path.join(OUT, name + ".png"), s, 8, 2)
    print("wrote 4 PNGs to", OUT)


if __name__ == "__main__":
    main()
