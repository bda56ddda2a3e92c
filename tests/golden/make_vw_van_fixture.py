#!/usr/bin/env python3
"""Makes tests/golden/scenes/vw-van from /root/reference/Scenes/vw-van (BASELINE.json configs[3]; run in the build container, where the
reference tree is present): copies the 161 PLY meshes that exist, and the scene file with its two absent inputs edited out -- the Shape
that names geometry/mesh_00125.ply and the environment map, replaced by the Teapot fixture's synthetic sky (both listed in /root/reference/.MISSING_LARGE_BLOBS) -- then pins the build's
loader against the REFERENCE PARSER's reading of that file: per-record sha256 digests written by oracle/_ref/pbrt_dump
(tests/golden/vw-van.parser.digest.json, compared in tests/test_host_scene.py)."""
import hashlib, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/Scenes/vw-van"
DST = os.path.join(ROOT, "tests", "golden", "scenes", "vw-van")


def main():
    os.makedirs(os.path.join(DST, "geometry"), exist_ok=True)
    for f in sorted(os.listdir(os.path.join(REF, "geometry"))):
        shutil.copy2(os.path.join(REF, "geometry", f), os.path.join(DST, "geometry", f))
    out, removed = [], 0
    for ln in open(os.path.join(REF, "vw-van.pbrt")).read().split("\n"):
        if "geometry/mesh_00125.ply" in ln:
            out.append('                # Shape "plymesh" "string filename" "geometry/mesh_00125.ply"   -- absent from the reference tree (.MISSING_LARGE_BLOBS): the body shell is left out'); removed += 1
        elif 'LightSource "infinite"' in ln and "pisa_latlong" in ln:
            out.append('  LightSource "infinite"  "string mapname" "textures/sky.hdr"   # textures/pisa_latlong.hdr is absent from the reference tree (.MISSING_LARGE_BLOBS): the synthetic sky of the Teapot fixture (oracle/make_golden.py)')
        else:
            out.append(ln)
    assert removed == 1
    os.makedirs(os.path.join(DST, "textures"), exist_ok=True)
    shutil.copy2(os.path.join(ROOT, "tests", "golden", "scenes", "Teapot", "textures", "sky.hdr"), os.path.join(DST, "textures", "sky.hdr"))   # TracerBoy.cpp:1903-1906 always loads mapName: a light without a map would leave the scene black
    header = ("# tests/golden/scenes/vw-van/vw-van.pbrt -- /root/reference/Scenes/vw-van/vw-van.pbrt (BASELINE.json configs[3]) with the two things its tree lacks\n"
              "# edited out: the Shape of geometry/mesh_00125.ply (682 837 triangles, the body shell) and the environment map (a synthetic sky instead).  161 PLY meshes, 240 ObjectInstances,\n"
              "# glass / metal / uber / mix materials: the rest of the file is the reference's, byte for byte (made by tests/golden/make_vw_van_fixture.py).\n")
    open(os.path.join(DST, "vw-van.pbrt"), "w").write(header + "\n".join(out))
    # the reference parser's own reading of the fixture
    dump = os.path.join(ROOT, "oracle", "_ref", "pbrt_dump")
    if not os.path.exists(dump): subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_golden
    make_golden.write_digest(os.path.join(DST, "vw-van.pbrt"), os.path.join(ROOT, "tests", "golden", "vw-van.parser.digest.json"))


if __name__ == "__main__":
    main()
