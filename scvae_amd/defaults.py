"""Default settings (``scvae/defaults.py:19-24``): one JSON document with the
same keys and values as the reference's ``scvae/defaults.json`` (data, not code)."""

import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                       "defaults.json")) as _file:
    defaults = json.load(_file)
