"""Run artefacts in the reference's shapes: per-episode reward CSVs
(``workers/trainer.py:552-568, 598-611``; column names ``src/env/env.py:5-11``; file names
``src/config.py:135-138``), ``conf.json`` (``src/util.py:28-31``) and per-agent checkpoints whose tensors are in
Keras ``model.weights`` order under the reference's file stems (``trainer.py:581-594``: ``actor1_1`` ...), as
``.npz`` (h5py is not available here), so an ``esim``-style reload of the actors is possible."""
import csv
import json
import os

import numpy as np

PLATOON_COL, SEED_COL, EPISODIC_REWARD_AVGWINDOW_COL, VEHICLE_COL = "platoon", "seed", "avg window", "Vehicle %s"
FNAME = {"actor": "actor%s_%s", "critic": "critic%s_%s", "target_actor": "target_actor%s_%s",
         "target_critic": "target_critic%s_%s"}


def reward_rows(pl_reward_lists, seed, platoon_tag, avg_window=None):
    """One platoon's table: row index = episode, one 'Vehicle m' column per vehicle, then seed, platoon
    (and 'avg window' for the trailing-mean table) -- the DataFrame of trainer.py:598-611."""
    M = len(pl_reward_lists)
    header = [""] + [VEHICLE_COL % (m + 1) for m in range(M)] + [SEED_COL, PLATOON_COL]
    if avg_window is not None:
        header.append(EPISODIC_REWARD_AVGWINDOW_COL)
    rows = []
    for ep in range(len(pl_reward_lists[0])):
        row = [ep] + [float(pl_reward_lists[m][ep]) for m in range(M)] + [seed, platoon_tag]
        if avg_window is not None:
            row.append(avg_window)
        rows.append(row)
    return header, rows


def generate_csvs(base_dir, conf, all_ep_reward_lists, all_avg_reward_lists):
    """trainer.py:552-568: platoon tables appended one under another (each restarting its episode index)."""
    paths = {}
    for name, lists, win in (("avg_ep_reward__seed%s.csv", all_avg_reward_lists, conf.reward_averaging_window),
                             ("ep_reward__seed%s.csv", all_ep_reward_lists, None)):
        path = os.path.join(base_dir, name % conf.random_seed)
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            for p, pl in enumerate(lists):
                header, rows = reward_rows(pl, conf.random_seed, p + 1, win)
                if p == 0:
                    w.writerow(header)
                w.writerows(rows)
        paths[name.split("__")[0]] = path
    return paths


def config_writer(path, conf):
    """src/util.py:28-31: the Config's __dict__ as JSON."""
    with open(path, "w") as f:
        json.dump({k: v for k, v in conf.__dict__.items() if isinstance(v, (int, float, str, bool, list, type(None)))}, f)


def config_loader(path, conf_cls):
    conf = conf_cls()
    for k, v in json.load(open(path)).items():
        if hasattr(conf, k):
            setattr(conf, k, v)
    return conf.refresh()


def save_agents(base_dir, agents, P, M, shared=False):
    """actor/critic/target weights of every (platoon, vehicle) agent, Keras `.weights` order."""
    for p in range(P):
        for m in range(M):
            k = m if shared else p * M + m
            for which, target, stem in (("actor", False, "actor"), ("critic", False, "critic"),
                                        ("actor", True, "target_actor"), ("critic", True, "target_critic")):
                np.savez(os.path.join(base_dir, FNAME[stem] % (p + 1, m + 1) + ".npz"),
                         *agents.get_weights(k, which, target=target))


def load_actor_weights(base_dir, pl_idx, m):
    """The 14 actor tensors of vehicle m (1-based) of platoon pl_idx, as saved by save_agents."""
    z = np.load(os.path.join(base_dir, FNAME["actor"] % (pl_idx, m) + ".npz"))
    return [z[f"arr_{i}"] for i in range(len(z.files))]
