"""LeRobot plugin entry point: `--policy.discover_packages_path=vla_fastvlm.lerobot_fastvla --policy.type=fastvla`
(reference: src/vla_fastvlm/lerobot_fastvla/__init__.py:3-11)."""
from .configuration_fastvla import FastVLAConfig
from .modeling_fastvla import FastVLAPolicy
from .processor_fastvla import make_fastvla_pre_post_processors

__all__ = ["FastVLAConfig", "FastVLAPolicy", "make_fastvla_pre_post_processors"]
