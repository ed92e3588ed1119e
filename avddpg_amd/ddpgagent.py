"""``policy`` and ``update_target`` with the reference's signatures (agent/ddpgagent.py:6-55)."""
import numpy as np
import torch

from ._hip import call, ptr, stream_handle


def policy(actor_state, noise_object=None, lbound=None, hbound=None):
    """ddpgagent.py:6-29: squeeze the actor output, add OU noise, clip; returns ``[np scalar]``."""
    out = torch.as_tensor(actor_state, dtype=torch.float32).reshape(-1).cuda().contiguous()
    noise = None
    if noise_object is not None:
        noise = torch.as_tensor(np.asarray(noise_object(), dtype=np.float32).reshape(-1)).cuda()
    action = torch.empty_like(out)
    call("avd_policy_f32", out.numel(), ptr(out), ptr(noise), float(lbound), float(hbound), ptr(action),
         stream_handle())
    return [np.squeeze(action.cpu().numpy().astype(np.float64))]


def _mix(tau, weights, targets):
    flat_w = torch.cat([torch.as_tensor(np.asarray(w, dtype=np.float32)).reshape(-1) for w in weights]).cuda()
    flat_t = torch.cat([torch.as_tensor(np.asarray(t, dtype=np.float32)).reshape(-1) for t in targets]).cuda()
    call("avd_polyak_f32", flat_w.numel(), ptr(flat_w), ptr(flat_t), float(tau), stream_handle())
    out, host, at = [], flat_t.cpu().numpy(), 0
    for t in targets:
        n = int(np.prod(np.shape(t)))
        out.append(host[at:at + n].reshape(np.shape(t)))
        at += n
    return out


def update_target(tau, t_critic_weights, critic_weights, t_actor_weights, actor_weights):
    """ddpgagent.py:31-55 -- pure: returns (tc_new_weights, ta_new_weights); the caller assigns."""
    return _mix(tau, critic_weights, t_critic_weights), _mix(tau, actor_weights, t_actor_weights)
