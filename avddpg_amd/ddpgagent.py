"""``policy`` and ``update_target`` with the reference's signatures (agent/ddpgagent.py:6-55), plus the fused equivalents the
north star names: ``act`` (the per-model ``policy(actor(state), noise, lo, hi)`` loop of workers/trainer.py:286-289 as one actor
launch + one clip launch) and ``learn`` (``Trainer.learn``, workers/trainer.py:472-508, on reference-shaped objects)."""
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


def act(actor_models, states, noise_objects=None, lbound=None, hbound=None):
    """The fused form of ``[policy(actor_m(state_m), noise_m, lbound, hbound)[0] for m in models]`` (workers/trainer.py:286-289):
    ONE actor launch over all models (``avd_actor_forward_f32``: model m reads row m of ``states``) and one noise-add + clip launch
    (``avd_policy_f32``). ``actor_models``: a list of ``model.get_actor`` objects, one per row (the same object may repeat);
    ``noise_objects``: None or one ``OUActionNoise`` per row, called in row order -- the reference's draw order. Returns the clipped
    actions as a float64 array [n] (A = 1) / [n, A]. Same values as the loop, launch for launch the batch-1 kernels with n rows."""
    from . import _hip

    n = len(actor_models)
    lay = actor_models[0].lay
    x = torch.as_tensor(np.asarray([np.asarray(s_, dtype=np.float32).reshape(-1)[:lay.S] for s_ in states], dtype=np.float32)).cuda().contiguous()
    if x.shape != (n, lay.S):
        raise _hip.AvdError(f"act: {n} models need states [{n}, {lay.S}], got {tuple(x.shape)}")
    same = all(m_ is actor_models[0] for m_ in actor_models)
    theta = actor_models[0].theta if same else torch.cat([m_.theta for m_ in actor_models]).contiguous()
    stats = actor_models[0].stats if same else torch.cat([m_.stats for m_ in actor_models]).contiguous()
    out = torch.empty(n, lay.A, dtype=torch.float32, device="cuda")
    # set_mod: 1 = every row uses set 0 (one shared model); 0 = row v uses set v
    call("avd_actor_forward_f32", _hip.C.byref(lay), n, 1 if same else 0, ptr(theta), ptr(stats), ptr(x), lay.S,
         float(actor_models[0].high), ptr(out), stream_handle())
    noise = None
    if noise_objects is not None:
        nz = np.stack([np.broadcast_to(np.asarray(o(), dtype=np.float32).reshape(-1), (lay.A,)) for o in noise_objects])
        noise = torch.as_tensor(np.ascontiguousarray(nz)).cuda()
    action = torch.empty_like(out)
    call("avd_policy_f32", out.numel(), ptr(out), ptr(noise), float(lbound), float(hbound), ptr(action), stream_handle())
    a = action.cpu().numpy().astype(np.float64)
    return a[:, 0] if lay.A == 1 else a


def learn(rbuffer, actor_model, critic_model, target_actor, target_critic, gamma=0.99):
    """``Trainer.learn`` (workers/trainer.py:472-508) on reference-shaped objects: (critic_grad[14], actor_grad[10]) in
    ``trainable_variables`` order from one sampled batch, pure w.r.t. the weights (``avd_learn_f32``)."""
    from .trainer import learn as _learn
    return _learn(rbuffer, actor_model, critic_model, target_actor, target_critic, gamma)
