"""CPU checks of the oracle's federated loop variants (oracle/trainer.py): invariants that follow from the
reference's control flow (workers/trainer.py:304-456, 631-695) -- run on tiny networks so the whole file takes seconds."""
import numpy as np

from oracle import platoon, trainer


def _mk(**kw):
    base = dict(num_platoons=3, pl_size=2, seed=3, buffer_size=64, batch_size=4, H1=16, H2=8, Ha=4, steps_per_episode=12)
    base.update(kw)
    return trainer.RefTrainer(platoon.EnvParams(), **base)


def _flat(ws):
    return np.concatenate([np.ravel(w) for w in ws])


def test_weights_aggregation_writes_group0_average_into_every_model_and_target():
    """:433-456: `get_avg_params(...)[0]` -- for interfrl that is vehicle 0's average over the platoons."""
    tr = _mk(fed_method="interfrl", aggregation_method="weights", fed_update_delay_steps=3)
    tr.reset_episode()
    for i in range(5):  # adds 1..5: first learn at i = 4 (5th add > batch 4); i = 4 is not a multiple of 3 -> local update
        tr.step(0, i)
    assert tr.updates == 3 * 2
    a00, a10 = _flat(tr.actors[0][0]), _flat(tr.actors[1][0])
    assert not np.array_equal(a00, a10)  # local updates made the agents diverge
    before = [np.mean([tr.actors[p][0][k] for p in range(3)], axis=0) for k in range(len(tr.actors[0][0]))]
    tr.step(0, 5)  # i = 5: not valid either
    tr2_before = [np.mean([tr.actors[p][0][k] for p in range(3)], axis=0, dtype=np.float32) for k in range(len(tr.actors[0][0]))]
    tr.step(0, 6)  # i = 6: federated weights step, no local update on it
    for p in range(3):
        for m in range(2):
            for got, want in zip(tr.actors[p][m], tr2_before):
                assert np.allclose(got, want, rtol=1e-6, atol=1e-8)
            assert np.array_equal(_flat(tr.t_actors[p][m]), _flat(tr.actors[p][m]))
            assert np.array_equal(_flat(tr.t_critics[p][m]), _flat(tr.critics[p][m]))
    assert not np.allclose(_flat(before), _flat(tr2_before))


def test_intrafrl_directional_leaves_the_lead_vehicle_untouched_on_federated_steps():
    tr = _mk(fed_method="intrafrl", intra_directional_averaging=True)
    tr.reset_episode()
    a_lead0 = _flat(tr.actors[1][0]).copy()
    for i in range(8):
        tr.step(0, i)
    assert tr.updates > 0
    assert np.array_equal(_flat(tr.actors[1][0]), a_lead0) and tr.a_opts[1][0].t == 0  # never stepped (:417-418)
    assert not np.array_equal(_flat(tr.actors[1][1]), a_lead0) and tr.a_opts[1][1].t == tr.updates // 6
    # without the flag the lead vehicle follows its platoon's mean gradient like the others
    tr = _mk(fed_method="intrafrl", intra_directional_averaging=False, pl_size=3)
    tr.reset_episode()
    for i in range(8):
        tr.step(0, i)
    assert np.array_equal(_flat(tr.actors[2][0]), _flat(tr.actors[2][1])) and np.array_equal(_flat(tr.actors[2][0]), _flat(tr.actors[2][2]))
    assert not np.array_equal(_flat(tr.actors[2][0]), _flat(tr.actors[0][0]))  # platoons differ


def test_schedule_quirk_valid_step_of_a_non_update_episode_updates_nothing():
    """fed_update_count = 2, delay 1: every step is a "valid update step", so the local-update gate (:345) is closed in ALL
    episodes, and in odd episodes the federated branch is closed too (:680): parameters stand still."""
    tr = _mk(fed_method="interfrl", fed_update_count=2, steps_per_episode=7)
    tr.run(1)  # episode 0: federated updates from the 5th add on
    assert tr.a_opts[0][0].t == 3
    snap = _flat(tr.actors[0][0]).copy()
    tr.reset_episode()
    for i in range(7):
        tr.step(1, i)  # episode 1
    assert tr.a_opts[0][0].t == 3 and np.array_equal(_flat(tr.actors[0][0]), snap) and tr.updates == 6 * 3 + 6 * 7
    tr.update_reward_list()
    tr.reset_episode()
    tr.step(2, 0)
    assert tr.a_opts[0][0].t == 4


def test_weighted_frl_starts_at_the_window_and_uses_inverse_mean_reward_weights():
    tr = _mk(fed_method="interfrl", weighted_average_enabled=True, weighted_window=2, steps_per_episode=6)
    tr.run(2)
    assert tr.fed_weight_sums is None  # episodes 0, 1 < window: unweighted mean
    tr.reset_episode()
    tr.step(2, 0)
    want = [[abs(1 / np.mean(tr.all_ep_reward_lists[p][m][-2:])) for p in range(3)] for m in range(2)]
    assert np.allclose(tr.fed_weights, want, rtol=1e-6)
    assert np.allclose(tr.fed_weight_sums, np.sum(want, axis=1), rtol=1e-6)
    # the P copies of vehicle m still receive the same (weighted) average: they stay identical
    assert np.array_equal(_flat(tr.actors[0][1]), _flat(tr.actors[2][1]))
