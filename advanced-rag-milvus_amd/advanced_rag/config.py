"""YAML -> PipelineConfig (reference config.py:18-32)."""
from __future__ import annotations

from typing import Any, Dict

from .pipeline import PipelineConfig


def load_yaml_config(path: str) -> Dict[str, Any]:
    import yaml
    with open(path, "r", encoding="utf-8") as f:
        return yaml.safe_load(f) or {}


def load_pipeline_config(path: str) -> PipelineConfig:
    return PipelineConfig(**((load_yaml_config(path).get("pipeline") or {})))
