"""The inference config of the reference (`dad_3dnet.yaml:1-12`) as a dict, plus a YAML loader (`utils.py:10-12`)."""
from __future__ import annotations

from typing import Any, Dict

DAD_3DNET_CONFIG: Dict[str, Any] = {
    "model_path": ".dad_checkpoints/dad_3dheads.trcd",
    "stride": 4,
    "img_size": 256,
    # insertion order == FlameParams.from_3dmm slicing order (predictor.find_3dmm_idx walks this dict)
    "constants": {"shape": 300, "expression": 100, "jaw": 3, "rotation": 6, "eyeballs": 0, "neck": 0,
                  "translation": 3, "scale": 1},
}


def load_default_config() -> Dict[str, Any]:
    return {**DAD_3DNET_CONFIG, "constants": dict(DAD_3DNET_CONFIG["constants"])}


def load_yaml(path: str) -> Dict[str, Any]:
    import yaml

    with open(path) as fd:
        return yaml.load(fd, yaml.FullLoader)
