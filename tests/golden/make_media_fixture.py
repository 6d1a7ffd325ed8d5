"""Collects the reference's own recipes (media/*.yaml: lists of images / videos, each `update_attributes` on top of
default-config.yaml + a scene file) into tests/golden/media_recipes.json -- DATA: parameter overrides and scene descriptions,
de-duplicated.  Run in the build container (reads /root/reference); the GPU box only sees the JSON.
    python tests/golden/make_media_fixture.py"""
import glob, json, os, sys
import yaml
MEDIA = "/root/reference/media"
out, seen = [], set()
for path in sorted(glob.glob(MEDIA + "/*.yaml")):
    doc = yaml.safe_load(open(path))
    if not isinstance(doc, list):
        continue            # a scene file
    for k, e in enumerate(doc):
        if not isinstance(e, dict) or "update_attributes" not in e and "scene_file" not in e:
            continue
        scene_file = e.get("scene_file", "../default-scene.yaml")
        scene = yaml.safe_load(open(os.path.normpath(os.path.join(MEDIA, scene_file))))
        attrs = e.get("update_attributes") or {}
        key = json.dumps([attrs, scene], sort_keys=True)
        if key in seen:
            continue
        seen.add(key)
        out.append({"recipe": f"{os.path.basename(path)}#{k}", "scene_file": os.path.basename(scene_file), "time": e.get("time"),
                    "update_attributes": attrs, "scene": scene})
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "media_recipes.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(len(out), "recipes ->", dst)
