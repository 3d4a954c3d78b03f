"""Compile the reference's TidyBot MJCF scenes into the JSON model tables shipped
under mink_amd/data/.  Runs only where /root/reference is mounted; the GPU box
loads the committed JSON tables instead."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from mink_amd.mjcf import load_mjcf  # noqa: E402

REF = os.environ.get("MINK_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "interactive_scripts", "stanford_tidybot")
OUT = os.path.join(os.path.dirname(__file__), "..", "mink_amd", "data")

for xml, name in [("cube.xml", "tidybot_cube"), ("open.xml", "tidybot_open"),
                  ("scene_mobile_kinova.xml", "mobile_kinova")]:
    m = load_mjcf(os.path.join(SRC, xml))
    m.save_json(os.path.join(OUT, name + ".json"))
    print(name, "nq", m.nq, "nv", m.nv, "nbody", m.nbody, "njnt", m.njnt,
          "ngeom", m.ngeom, "nsite", m.nsite, "nkey", m.nkey)
