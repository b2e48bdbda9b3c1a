"""The constants table: ``config`` is the parsed yolo_amd/config.yml (anchors, strides, xyscale,
anchors per scale, IoU loss threshold), the same data the reference reads at import
(/root/reference/config.py:1-9, /root/reference/config.yml:1-7)."""
import pathlib

import yaml

config = yaml.safe_load((pathlib.Path(__file__).parent / "config.yml").read_text())
