"""LeRobot pre/post-processor pipelines for FastVLA (reference:
src/vla_fastvlm/lerobot_fastvla/processor_fastvla.py:22-61).  The steps are LeRobot-owned elementwise
normalisation (out of scope for the HIP path); this module only assembles them, and needs `lerobot` installed."""
from __future__ import annotations

from typing import Any

import torch

from ._lerobot_compat import HAVE_LEROBOT
from .configuration_fastvla import FastVLAConfig


def make_fastvla_pre_post_processors(config: FastVLAConfig, dataset_stats: dict[str, dict[str, torch.Tensor]] | None = None,
                                     fold_into=None):
    """`fold_into` (an extension of this build): a FastVLAPolicy that takes the STATE normalisation and the
    ACTION un-normalisation into its head kernels (`policy.fold_dataset_stats`).  The pipelines then keep only what is left for
    LeRobot to do: the pre-processor still normalises the ACTION targets of training batches (the loss lives in normalised
    space; VISUAL is IDENTITY in this policy's map), the post-processor only moves the action to the CPU."""
    if not HAVE_LEROBOT:
        raise ModuleNotFoundError("make_fastvla_pre_post_processors needs the `lerobot` package (>=0.4.4)")
    from lerobot.processor import (AddBatchDimensionProcessorStep, DeviceProcessorStep, NormalizerProcessorStep,
                                   PolicyAction, PolicyProcessorPipeline, RenameObservationsProcessorStep,
                                   UnnormalizerProcessorStep)
    from lerobot.processor.converters import policy_action_to_transition, transition_to_policy_action
    from lerobot.utils.constants import POLICY_POSTPROCESSOR_DEFAULT_NAME, POLICY_PREPROCESSOR_DEFAULT_NAME

    features = {**config.input_features, **config.output_features}
    if fold_into is not None and dataset_stats is not None:
        fold_into.fold_dataset_stats(dataset_stats)
        features = dict(config.output_features)   # targets only; STATE is normalised inside fv_head_forward now
    pre = PolicyProcessorPipeline[dict[str, Any], dict[str, Any]](
        steps=[RenameObservationsProcessorStep(rename_map={}), AddBatchDimensionProcessorStep(),
               DeviceProcessorStep(device=config.device),
               NormalizerProcessorStep(features=features, norm_map=config.normalization_mapping, stats=dataset_stats,
                                       device=config.device)],
        name=POLICY_PREPROCESSOR_DEFAULT_NAME)
    post_steps = [DeviceProcessorStep(device="cpu")]
    if fold_into is None or dataset_stats is None:
        post_steps.insert(0, UnnormalizerProcessorStep(features=config.output_features, norm_map=config.normalization_mapping,
                                                      stats=dataset_stats))
    post = PolicyProcessorPipeline[PolicyAction, PolicyAction](
        steps=post_steps,
        name=POLICY_POSTPROCESSOR_DEFAULT_NAME, to_transition=policy_action_to_transition,
        to_output=transition_to_policy_action)
    return pre, post
