"""Data helpers of the three case studies (reference examples/{localization,curvefit,gol}/data.py): the
blinker as a known-answer pattern of the Life rule and of the Gibbs inversion, the curvefit datasets (key
chains of jax.random draws reproduced by the library's Threefry: the normals are checked against the CPU
restatement's bit-compatible draws), the localization trajectories."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import genjax_amd

    assert torch.cuda.is_available()
    return genjax_amd


def test_blinker_known_answers(G, oracle):
    """examples/gol/data.py:6-38.  A horizontal blinker steps to the vertical one and back (period 2); the
    clipped corner pattern of get_blinker_n dies; the sampler inverts the blinker target exactly."""
    from genjax_amd.families import gol

    b4 = gol.get_blinker_4x4()
    assert b4.cpu().numpy().tolist() == [[0, 0, 0, 0], [0, 1, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]]
    v4 = gol.generate_next_state(b4)
    assert v4.cpu().numpy().tolist() == [[0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 0, 0]]
    assert torch.equal(gol.generate_next_state(v4), b4)
    b10 = gol.get_blinker_10x10()
    assert int(b10.sum()) == 3 and b10[2, 4:7].cpu().numpy().tolist() == [1, 1, 1]
    v10 = gol.generate_next_state(b10)
    assert v10[1:4, 5].cpu().numpy().tolist() == [1, 1, 1] and int(v10.sum()) == 3
    assert torch.equal(gol.generate_next_state(v10), b10)
    np.testing.assert_array_equal(v10.cpu().numpy(), oracle.gol_step(b10.cpu().numpy()))
    bn = gol.get_blinker_n(6)
    assert int(bn.sum()) == 2 and bn[4, 4:6].cpu().numpy().tolist() == [1, 1]          # the slice is clipped
    assert int(gol.generate_next_state(bn).sum()) == 0
    # inversion: the chains start from random boards (about half of the 100 successor bits wrong) and settle on
    # posterior samples, whose successor differs from the blinker in about p_flip * 100 = 3 cells
    res = gol.run_sampler_and_get_summary(oracle.key(1), gol.GibbsSampler(b10, 0.03), 300, 100, n_chains=32)
    assert res.n_incorrect_bits.shape == (4, 32)
    assert float(res.n_incorrect_bits[0].float().mean()) > 25
    assert float(res.n_incorrect_bits[-1].float().mean()) < 5 and int(res.n_incorrect_bits[-1].max()) <= 10
    assert bool((res.predictive_posterior_scores[-1] > res.predictive_posterior_scores[0]).all())


def test_curvefit_datasets(G, oracle):
    """examples/curvefit/data.py:30-306: shapes, the key-chain draws (bit-compatible normals), the planted
    outliers, the reference dataset's fixed coefficients."""
    from genjax_amd.families import curvefit_data as D

    ref = D.get_reference_dataset()
    assert ref["xs"].shape == (20,) and ref["true_params"] == {"a": -0.211, "b": -0.395, "c": 0.673}
    # key = key(42); key, subkey = split(key); noise = 0.05 * normal(subkey, (20,))
    sub = oracle.split(oracle.key(42), 2)[1]
    noise = np.float32(0.05) * oracle.normal(sub, 20)
    xs = np.linspace(0, 1, 20, dtype=np.float32)
    want = (np.float32(-0.211) + np.float32(-0.395) * xs + np.float32(0.673) * xs * xs) + noise
    np.testing.assert_allclose(ref["ys"].cpu().numpy(), want, rtol=2e-6, atol=2e-7)
    again = D.generate_fixed_dataset(n_points=20)
    assert torch.equal(again["ys"], ref["ys"])

    easy = D.generate_easy_inference_dataset()
    assert easy["xs"].shape == (5,) and easy["n_points"] == 5
    k = oracle.key(42)
    coeffs = []
    for _ in range(3):
        k, sub = oracle.split(k, 2)
        coeffs.append(np.float32(0.3) * oracle.normal(sub, 1)[0])
    for name, c in zip("abc", coeffs):
        np.testing.assert_allclose(easy["true_params"][name], c, rtol=1e-5)

    out = D.generate_dataset_with_outliers(n_points=2000, outlier_fraction=0.15, true_a=0.1, true_b=0.2, true_c=0.3)
    frac = float(out["is_outlier"].float().mean())
    assert abs(frac - 0.15) < 0.03 and out["n_outliers"] == int(out["is_outlier"].sum())
    resid = (out["ys"] - out["clean_ys"]).double()
    assert abs(float(resid[~out["is_outlier"]].std()) - 0.05) < 0.005
    assert abs(float(resid[out["is_outlier"]].std()) - 0.15) < 0.02

    test = D.generate_test_dataset(n_points=10, seed=7)
    assert test["ys"].shape == (10,) and set(test["true_params"]) == {"a", "b", "c", "noise_std"}
    assert float((test["ys"] - test["clean_ys"]).abs().max()) < 0.3          # observation noise 0.05
    D.print_dataset_summary(out, "outliers")


def test_localization_trajectories(G, oracle):
    """examples/localization/data.py:104-244: the exploration waypoints stay inside room 1; both trajectory types
    produce one 8-ray observation per waypoint; several trajectories cycle through the types under split keys."""
    from genjax_amd.families import localization as L

    world = L.create_basic_multi_room_world()
    wp = L.create_waypoint_trajectory_exploration_room1()
    assert len(wp) == 10 and all(0 < x < 4 and 0 < y < 10 for x, y in wp)
    poses, controls, obs = L.generate_ground_truth_data(world, oracle.key(3), trajectory_type="exploration")
    assert len(poses) == 10 and len(controls) == 9 and len(obs) == 10 and obs[0].shape == (8,)
    assert all(np.all(o >= 0) and np.all(o <= 10.5) for o in obs)
    with pytest.raises(ValueError, match="Unknown trajectory_type"):
        L.generate_ground_truth_data(world, oracle.key(3), trajectory_type="spiral")
    trajs = L.generate_multiple_trajectories(world, oracle.key(4), n_trajectories=3)
    assert [t for t, _ in trajs] == ["room_navigation", "exploration", "room_navigation"]
    assert len(trajs[0][1][0]) == 17 and len(trajs[1][1][0]) == 10
    # independent keys: the two room-navigation records carry different noise
    assert not np.allclose(trajs[0][1][2][0], trajs[2][1][2][0])
