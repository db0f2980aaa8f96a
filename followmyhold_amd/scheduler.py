"""Flow-matching Euler scheduler with reversed time (sigmas run 0 -> 1), host-side mirror of the reference's
patched scheduler (third_party_patches/hy3dgen/shapegen/schedulers.py; SCH below).

Only the members the guidance loop touches are provided: set_timesteps (SCH:171-211), step (SCH:235-318) and
step_final (SCH:411-493), with the reference's behaviour kept as is:
  * `sigmas` gets a trailing 1.0 appended, `timesteps = sigmas * num_train_timesteps`;
  * the first `step()` initialises the index with index_for_timestep(), which picks the SECOND match when a
    timestep occurs more than once (SCH:213-226);
  * `step_final` does not advance the index, so called after `step()` it reads the NEXT sigma
    (pipelines.py:1612 then :1621), while inside the inner loops it reads the current one (pipelines.py:1507);
  * arithmetic is done in float32 and cast back to the dtype of `model_output` (fp16 latents).
The update itself is a 196 608-element axpy that stays in PyTorch (negligible).
"""
from dataclasses import dataclass
from typing import List, Optional, Union

import numpy as np
import torch


@dataclass
class FlowMatchEulerDiscreteSchedulerOutput:
    prev_sample: torch.Tensor
    pred_x1: torch.Tensor


class _Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting: bool = False):
        self.config = _Config(num_train_timesteps=num_train_timesteps, shift=shift,
                              use_dynamic_shifting=use_dynamic_shifting)
        # training-time table: sigma_k = k / N for k = 1..N (exact integers in float32), statically shifted
        table = torch.arange(1, num_train_timesteps + 1, dtype=torch.float32) / num_train_timesteps
        if not use_dynamic_shifting:
            table = self._static_shift(table)
        self.sigmas = table.cpu()
        self.timesteps = table * num_train_timesteps
        self.sigma_min, self.sigma_max = self.sigmas[-1].item(), self.sigmas[0].item()
        self._step_index = self._begin_index = None

    def _static_shift(self, s):
        k = self.config.shift
        return k * s / (1 + (k - 1) * s)

    step_index = property(lambda self: self._step_index)
    begin_index = property(lambda self: self._begin_index)

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps: int = None, device: Union[str, torch.device] = None,
                      sigmas: Optional[List[float]] = None, mu: Optional[float] = None):
        """Inference schedule.  The guidance pipeline passes sigmas = linspace(0, 1, 20) (pipelines.py:1187-1193)."""
        dynamic = self.config.use_dynamic_shifting
        if dynamic and mu is None:
            raise ValueError("set_timesteps: `mu` is required because this scheduler was configured with use_dynamic_shifting=True")
        n_train = self.config.num_train_timesteps
        if sigmas is None:
            self.num_inference_steps = num_inference_steps
            sigmas = np.linspace(self.sigma_max * n_train, self.sigma_min * n_train, num_inference_steps) / n_train
        sigmas = np.asarray(sigmas)
        sigmas = (np.exp(mu) / (np.exp(mu) + (1 / sigmas - 1) ** 1.0)) if dynamic else self._static_shift(sigmas)
        sig = torch.from_numpy(sigmas).to(dtype=torch.float32, device=device)
        self.timesteps = (sig * n_train).to(device=device)
        self.sigmas = torch.cat([sig, torch.ones(1, device=sig.device)])  # trailing 1.0 = clean sample
        self._step_index = self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        """Position of `timestep` in the schedule.  A value that occurs more than once resolves to its SECOND occurrence
        (so that a run started in the middle of a schedule does not skip a sigma, SCH:213-226); a value that does not
        occur raises IndexError like the reference's indexing does."""
        table = self.timesteps if schedule_timesteps is None else schedule_timesteps
        hits = torch.nonzero(table == timestep).flatten().tolist()
        return hits[1] if len(hits) > 1 else hits[0]

    def _locate(self, timestep):
        """First use after set_timesteps(): the index comes from set_begin_index() when the caller fixed it, otherwise
        from the timestep's place in the schedule."""
        if self._step_index is not None:
            return
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        if torch.is_tensor(timestep):
            timestep = timestep.to(self.timesteps.device)
        self._step_index = self.index_for_timestep(timestep)

    _init_step_index = _locate   # name used by callers written against the reference's scheduler

    @staticmethod
    def _reject_int_timestep(timestep):
        if isinstance(timestep, int) or isinstance(timestep, (torch.IntTensor, torch.LongTensor)):
            raise ValueError("step(): `timestep` must be one of scheduler.timesteps (a float tensor), not an integer index such as "
                             "the counter of enumerate(timesteps)")

    def _euler(self, model_output, sample, *targets):
        """x + (target - sigma_k) * v for every target sigma, evaluated in float32 and returned in v's dtype."""
        x = sample.to(torch.float32)
        here = self.sigmas[self._step_index]
        return [(x + (there - here) * model_output).to(model_output.dtype) for there in targets]

    def step(self, model_output, timestep, sample, return_dict: bool = True, **_):
        """One Euler step towards sigma_{k+1} plus the clean-sample prediction (target sigma 1); advances the index."""
        self._reject_int_timestep(timestep)
        self._locate(timestep)
        prev_sample, pred_x1 = self._euler(model_output, sample, self.sigmas[self._step_index + 1], 1)
        self._step_index += 1
        if return_dict:
            return FlowMatchEulerDiscreteSchedulerOutput(prev_sample=prev_sample, pred_x1=pred_x1)
        return (prev_sample, pred_x1)

    def step_final(self, model_output, timestep, sample, return_dict: bool = True, **_):
        """Clean-sample prediction only; the index stays where it is (the inner optimisation loops call this repeatedly)."""
        self._reject_int_timestep(timestep)
        self._locate(timestep)
        (pred_x1,) = self._euler(model_output, sample, 1)
        return pred_x1 if return_dict else (pred_x1,)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    """pipelines.py:363-419."""
    if timesteps is not None and sigmas is not None:
        raise ValueError("retrieve_timesteps: pass either `timesteps` or `sigmas`, not both")
    if timesteps is not None:
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
        timesteps = scheduler.timesteps
        num_inference_steps = len(timesteps)
    elif sigmas is not None:
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
        timesteps = scheduler.timesteps
        num_inference_steps = len(timesteps)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
        timesteps = scheduler.timesteps
    return timesteps, num_inference_steps
