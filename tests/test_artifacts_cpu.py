"""CPU tests of the run artefacts / CLI surface: reward CSV schema, conf.json round trip,
reference flag names and their override quirks. No GPU, no kernels."""
import csv
import os

import numpy as np

from avddpg_amd import artifacts
from avddpg_amd.__main__ import get_cmdl_args
from avddpg_amd.config import Config


def test_reward_csv_schema(tmp_path):
    conf = Config(num_platoons=2, pl_size=3)
    ep = [[[np.float32(-1.5 * (e + 1) * (m + 1) * (p + 1)) for e in range(4)] for m in range(3)] for p in range(2)]
    avg = [[[float(np.mean(ep[p][m][:e + 1])) for e in range(4)] for m in range(3)] for p in range(2)]
    paths = artifacts.generate_csvs(str(tmp_path), conf, ep, avg)
    assert os.path.basename(paths["ep_reward"]) == "ep_reward__seed1.csv"
    assert os.path.basename(paths["avg_ep_reward"]) == "avg_ep_reward__seed1.csv"
    rows = list(csv.reader(open(paths["ep_reward"])))
    assert rows[0] == ["", "Vehicle 1", "Vehicle 2", "Vehicle 3", "seed", "platoon"]
    assert len(rows) == 1 + 2 * 4 and rows[1][0] == "0" and rows[5][0] == "0" and rows[5][-1] == "2"  # index restarts per platoon
    assert float(rows[2][2]) == float(ep[0][1][1]) and rows[1][4] == "1"
    rows = list(csv.reader(open(paths["avg_ep_reward"])))
    assert rows[0][-1] == "avg window" and rows[1][-1] == "40"


def test_conf_json_roundtrip(tmp_path):
    conf = Config(pl_size=5, fed_method="interfrl", model="ModelA")
    p = str(tmp_path / "conf.json")
    artifacts.config_writer(p, conf)
    back = artifacts.config_loader(p, Config)
    assert (back.pl_size, back.fed_method, back.model, back.fed_enabled) == (5, "interfrl", "ModelA", True)
    assert back.steps_per_episode == 600


def test_cli_flag_names_and_override_quirks():
    args, conf = get_cmdl_args(["tr"], Config())
    # reference quirks (src/cmd/api.py:78, 81, 36-37, 43-44): CLI defaults win over Config defaults
    assert conf.weighted_average_enabled is False and Config().weighted_average_enabled is True
    assert conf.intra_directional_averaging is True and Config().intra_directional_averaging is False
    args, conf = get_cmdl_args(["tr", "--seed", "7", "--pl_num", "4096", "--pl_size", "5", "--fed_method", "interfrl",
                                "--fed_update_delay", "0.3", "--fed_agg_method", "weights", "--total_time_steps", "6000",
                                "--method", "exact", "--fed_weight_enabled", "x", "--rand_states", ""], Config())
    assert (conf.random_seed, conf.num_platoons, conf.pl_size, conf.fed_method) == (7, 4096, 5, "interfrl")
    # int(0.3 / 0.1) == 2 in floating point: the reference computes it the same way (src/cmd/api.py:41)
    assert conf.fed_update_delay_steps == 2 and conf.aggregation_method == "weights" and conf.number_of_episodes == 10
    assert conf.method == "exact" and conf.weighted_average_enabled is True and conf.rand_states is False
    args, _ = get_cmdl_args(["esim", "some/dir", "--n_timesteps", "50"], Config())
    assert args.exp_path == "some/dir" and args.n_timesteps == 50
