"""`python -m avddpg_amd {tr,esim}` -- the two modes of the reference CLI that touch the hot path, with the
reference's flag names and override quirks (``src/cmd/api.py:5-50, 59-88``): ``--fed_weight_enabled`` defaults to
False and always overrides the Config default (True); ``--intra_directional_averaging`` defaults to True; the
``type=bool`` flags treat any non-empty string as True. Reporting modes (accumr/accums/lsim/lmany/pid) are out of scope."""
import argparse
import datetime
import os
import sys

from .config import Config


def get_cmdl_args(argv, conf):
    ap = argparse.ArgumentParser(prog="python -m avddpg_amd", description="avddpg hot path on MI355X")
    sub = ap.add_subparsers(dest="mode")
    tr = sub.add_parser("tr", help="run in training mode")
    tr.add_argument("--seed", type=int, default=conf.random_seed)
    tr.add_argument("--method", choices=[conf.exact, conf.euler])
    tr.add_argument("--rand_states", type=bool)
    tr.add_argument("--total_time_steps", type=int)
    tr.add_argument("--pl_num", type=int)
    tr.add_argument("--pl_size", type=int)
    tr.add_argument("--buffer_size", type=int)
    tr.add_argument("--actor_lr", type=float)
    tr.add_argument("--critic_lr", type=float)
    tr.add_argument("--fed_method", choices=[conf.interfrl, conf.intrafrl, conf.nofrl])
    tr.add_argument("--fed_update_count", type=int)
    tr.add_argument("--fed_cutoff_ratio", type=float)
    tr.add_argument("--fed_update_delay", type=float)
    tr.add_argument("--fed_weight_enabled", type=bool, default=False)
    tr.add_argument("--fed_weight_window", type=int)
    tr.add_argument("--fed_agg_method", type=str, choices=["gradients", "weights"])
    tr.add_argument("--intra_directional_averaging", type=bool, default=True)
    tr.add_argument("--rng", choices=["host", "device"], default="host",
                    help="host: reference RNG stream (fixed-seed parity); device: Philox in the kernels (throughput)")
    tr.add_argument("--engine", choices=["per_agent", "batched", "fused", "fused3"], default=None,
                    help="interfrl with every step federated (shared weight sets): per_agent = exact f32 kernel per agent + federated "
                         "sum (default); fused3 = split-operand set learner, f32-class results, ~4x faster; fused / batched = bf16 "
                         "operands (not in the reference CLI)")
    tr.add_argument("--episodes", choices=["reference", "platoon"], default="reference",
                    help="reference: the episode loop of workers/trainer.py:232-273 on the host (any terminal platoon ends the episode of "
                         "all; per-episode reward CSVs in the reference's schema). platoon: THROUGHPUT mode (needs --rng device): "
                         "total_time_steps steps with every platoon running its own episodes on the device (avd_episode_end_f32), no "
                         "host synchronisation per step; writes curve.csv (episodes closed, mean episodic reward and length per "
                         "reporting window, evaluator score) instead of the per-episode CSVs (not in the reference CLI). The schedule "
                         "flags keep their meaning on an episode-EQUIVALENT clock (step // steps_per_episode: --fed_update_count, "
                         "--fed_cutoff_ratio); --fed_weight_enabled weights every agent by |1 / mean of its last fed_weight_window closed "
                         "episodes' rewards| from a device-side history, from step fed_weight_window x steps_per_episode on; conf.json "
                         "records the rule (episodes_mode, episode_clock)")
    tr.add_argument("--report_every", type=int, default=10000, help="--episodes platoon: steps per curve point")
    tr.add_argument("--save_platoons", type=int, default=None,
                    help="checkpoint the agents of the first N platoons only (default: all with --episodes reference, 4 with platoon)")
    tr.add_argument("--out", type=str, default=".outputs")
    es = sub.add_parser("esim", help="run in evaluation/simulator mode")
    es.add_argument("exp_path", type=str)
    es.add_argument("--n_timesteps", type=int, default=100)
    args = ap.parse_args(argv)
    if getattr(args, "save_platoons", None) is not None and args.save_platoons < 1:
        ap.error("--save_platoons must be >= 1 (esim reloads platoon 1's actors)")
    return args, set_args_to_config(args, conf)


def set_args_to_config(args, conf):
    """src/cmd/api.py:5-50."""
    g = lambda n: getattr(args, n, None)
    if g("seed") is not None:
        conf.random_seed = args.seed
    for flag, field in (("method", "method"), ("rand_states", "rand_states"), ("total_time_steps", "total_time_steps"),
                        ("pl_num", "num_platoons"), ("pl_size", "pl_size"), ("buffer_size", "buffer_size"),
                        ("actor_lr", "actor_lr"), ("critic_lr", "critic_lr"), ("fed_method", "fed_method"),
                        ("fed_update_count", "fed_update_count"), ("fed_cutoff_ratio", "fed_cutoff_ratio"),
                        ("intra_directional_averaging", "intra_directional_averaging"),
                        ("fed_update_delay", "fed_update_delay"), ("fed_weight_enabled", "weighted_average_enabled"),
                        ("fed_weight_window", "weighted_window"), ("fed_agg_method", "aggregation_method")):
        if g(flag) is not None:
            setattr(conf, field, getattr(args, flag))
    return conf.refresh()


def main(argv=None, conf=None):
    """conf: the Config the flags override (the reference edits src/config.py for anything the CLI has no flag for,
    e.g. ``framework``); default Config()."""
    conf = Config() if conf is None else conf
    args, conf = get_cmdl_args(sys.argv[1:] if argv is None else argv, conf)
    if args.mode == "tr":
        import numpy as np

        from . import artifacts, trainer
        np.random.seed(conf.random_seed)  # rand.set_global_seed (src/rand.py:6-15)
        base = os.path.join(args.out, datetime.datetime.now().strftime("%y%m%d_%H%M%S"))
        os.makedirs(base, exist_ok=True)
        if args.episodes == "platoon":
            if args.rng != "device":
                raise SystemExit("--episodes platoon needs --rng device")
            from . import evaluator
            nofrl = conf.fed_method == conf.nofrl
            vt = trainer.VecTrainer(conf, rng="device", auto_reset="platoon", shared_engine=args.engine, fused_update=nofrl)
            vt.reset_episode()
            rng_state = np.random.get_state()

            def score():  # workers/evaluator.py:145 on platoon 1's actors (the evaluator reseeds the global legacy RNG: put it back)
                grp = vt.agents
                if not vt.shared:
                    import copy
                    grp = copy.copy(vt.agents)
                    grp.theta, grp.stats, grp.n_sets = vt.agents.theta[:vt.M], vt.agents.stats[:vt.M], vt.M
                r = evaluator.run(conf=conf, actors=grp, pl_idx=1, set_mod=vt.M if vt.shared else 0)[0]
                np.random.set_state(rng_state)
                return float(r)

            with open(os.path.join(base, "curve.csv"), "w") as f:
                f.write("step,episodes_closed,mean_episodic_reward,mean_episode_length,evaluator_score\n")
                f.write(f"0,0,,,{score():.3f}\n")
                for k in range(1, conf.total_time_steps + 1):
                    vt.step()
                    if k % args.report_every == 0 or k == conf.total_time_steps:
                        r, ln, n = vt.env.pop_episode_stats()
                        f.write(f"{k},{n},{r:.5f},{ln:.2f},{score():.3f}\n")
                        f.flush()
            if vt.nonfinite_updates():
                print(f"warning: {vt.nonfinite_updates()} weight-set updates were skipped for non-finite gradients", file=sys.stderr)
        else:
            vt = trainer.VecTrainer(conf, rng=args.rng, auto_reset=False, shared_engine=args.engine)
            ep, avg = vt.run()
            artifacts.generate_csvs(base, conf, ep, avg)
        n_save = vt.P if args.save_platoons is None and args.episodes == "reference" else min(vt.P, 4 if args.save_platoons is None else args.save_platoons)
        artifacts.save_agents(base, vt.agents, n_save, vt.M, shared=vt.shared)
        # what ran, beside the reference's fields: how many platoons' agents the directory holds (esim loops over exactly these),
        # which episode rule applied and what the schedule predicates' `training_episode` was
        conf.saved_platoons = int(n_save)
        conf.episodes_mode = args.episodes
        conf.episode_clock = ("workers/trainer.py:232-273: one episode loop for all platoons, any terminal platoon ends it" if args.episodes == "reference"
                              else "per-platoon episodes on the device; schedule predicates on step // steps_per_episode; weighted averaging "
                                   "(if enabled) from step weighted_window x steps_per_episode on, weights from each agent's last "
                                   "weighted_window closed episodes")
        artifacts.config_writer(os.path.join(base, "conf.json"), conf)
        print(base)
    elif args.mode == "esim":
        from . import artifacts, evaluator, vec
        conf = artifacts.config_loader(os.path.join(args.exp_path, "conf.json"), Config)
        # model count and shapes as the trainer derives them (workers/evaluator.py:48-66): L models of S states / 1 action
        # decentralized, ONE model of 4L states / L actions and widths x1.2 centralized (src/environment.py:35-52)
        shape = vec.VecPlatoon(1, conf.pl_size, conf, rng="device")  # device RNG: consumes no np.random draws
        M = shape.num_models
        # a run saved with --save_platoons N holds the first N platoons' agents: conf.json says how many (older directories: all)
        import json
        saved = json.load(open(os.path.join(args.exp_path, "conf.json"))).get("saved_platoons", conf.num_platoons)
        for p in range(1, int(saved) + 1):
            if not os.path.exists(os.path.join(args.exp_path, artifacts.FNAME["actor"] % (p, 1) + ".npz")):
                raise FileNotFoundError(f"{args.exp_path}: no checkpoint of platoon {p}'s actors (conf.json records {saved} saved platoons)")
            grp = vec.AgentGroup(M, shape.num_states, shape.num_actions, conf, hidd_mult=shape.hidden_multiplier)
            for m in range(M):
                grp.set_weights(m, "actor", artifacts.load_actor_weights(args.exp_path, p, m + 1))
            rew, _ = evaluator.run(conf=conf, actors=grp, pl_idx=p, manual_timestep_override=args.n_timesteps)
            print(f"platoon {p}: cumulative platoon reward {rew}")
    else:
        raise SystemExit("modes: tr, esim")


if __name__ == "__main__":
    main()
