"""Seeded synthetic workloads for the likelihood hot path.

Input synthesis only: this module builds the *inputs* of the path (one-hot feature block,
cluster / confounder assignments, mixture weights, source assignment, Dirichlet prior
concentrations).  It computes nothing on the path itself -- feature counts, probability
tables and likelihoods come from the HIP engine (product), or from `oracle/` (tests).

Shapes follow BASELINE.json `configs`:
  cfg1      50 x 30 x 5,    K=2,  universal                      (plumbing)
  headline  1000 x 200 x 10, K=5,  universal                      (metric is quoted on this)
  stress    5000 x 500 x 20, K=10, universal + 2 x 20-group confounders
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

SHAPES = {
    #            N     F    S   K  extra confounders (n_groups each)   ragged states
    "cfg1":     (50,   30,  5,  2, (),                                  True),
    "headline": (1000, 200, 10, 5, (),                                  False),
    "stress":   (5000, 500, 20, 10, (20, 20),                           False),
}

NA_RATE = 0.03


@dataclass
class Workload:
    name: str
    features: np.ndarray                 # bool [N, F, S] one-hot, all-False row = NA (load_data.py:88,105)
    states_per_feature: np.ndarray       # bool [F, S] applicable states
    component_names: list                # ['clusters', 'universal', ...]
    groups: list                         # per component bool [G_c, N]; groups[0] = clusters
    concentration: list                  # per component float64: [F, S] for clusters, [G_c, F, S] for confounders
    weights: np.ndarray                  # float32 [F, C]   (state.py:546)
    source: np.ndarray                   # bool [N, F, C] one-hot over components, NA rows all False
    seeds: tuple = (0, 1)
    na_values: np.ndarray = field(init=False)

    def __post_init__(self):
        self.na_values = ~self.features.any(axis=-1)

    @property
    def shape(self):
        return self.features.shape

    @property
    def n_components(self):
        return len(self.groups)

    @property
    def clusters(self):
        return self.groups[0]


def _sample_one_hot(p, rng):
    """Draw one category per leading index from p[..., C] (inverse-cdf), one-hot bool out."""
    cdf = np.cumsum(p, axis=-1, dtype=np.float64)
    cdf /= cdf[..., -1:]
    z = rng.random(p.shape[:-1] + (1,))
    idx = np.argmax(z < cdf, axis=-1)
    return np.eye(p.shape[-1], dtype=bool)[idx]


def make_state(features, confounder_groups, n_clusters, seed):
    """Draw the sample-state part of a workload (clusters, weights, source) for `seed`.
    Used to build batches of distinct states over one resident feature block."""
    rng = np.random.default_rng(seed)
    n_objects, n_features, _ = features.shape
    a = rng.integers(0, 2 * n_clusters, size=n_objects)
    clusters = np.stack([a == k for k in range(n_clusters)])
    groups = [clusters, *confounder_groups]
    n_comp = len(groups)
    weights = rng.dirichlet(np.ones(n_comp), size=n_features).astype(np.float32)
    has_comp = np.stack([g.any(axis=0) for g in groups], axis=1)
    w = has_comp[:, None, :] * weights[None, :, :].astype(np.float64)
    w /= w.sum(axis=-1, keepdims=True)
    source = _sample_one_hot(w, rng)
    source[~features.any(axis=-1)] = False
    return clusters, weights, source


def make_workload(name: str = "headline", data_seed: int = 0, state_seed: int = 1,
                  shape=None) -> Workload:
    """Build the seeded synthetic workload `name` (or an explicit
    shape=(N, F, S, K, extra_confounder_groups, ragged))."""
    n_objects, n_features, n_states, n_clusters, extra, ragged = shape if shape is not None else SHAPES[name]
    rng = np.random.default_rng(data_seed)

    if ragged:
        n_states_f = rng.integers(2, n_states + 1, size=n_features)
    else:
        n_states_f = np.full(n_features, n_states)
    states_per_feature = np.arange(n_states)[None, :] < n_states_f[:, None]

    x = (rng.random((n_objects, n_features)) * n_states_f[None, :]).astype(np.int64)
    na = rng.random((n_objects, n_features)) < NA_RATE
    features = np.zeros((n_objects, n_features, n_states), dtype=bool)
    nn, ff = np.nonzero(~na)
    features[nn, ff, x[nn, ff]] = True

    component_names = ["clusters", "universal"]
    conf_groups = [np.ones((1, n_objects), dtype=bool)]
    for i, n_groups in enumerate(extra):
        g = rng.integers(0, n_groups + 1, size=n_objects)      # id n_groups = in no group
        conf_groups.append(np.stack([g == j for j in range(n_groups)]))
        component_names.append(f"conf{i + 1}")

    clusters, weights, source = make_state(features, conf_groups, n_clusters, state_seed)
    groups = [clusters, *conf_groups]

    unif = states_per_feature.astype(np.float64)               # uniform Dirichlet: 1.0 on applicable states
    concentration = [unif.copy()]
    for g in conf_groups:
        concentration.append(np.broadcast_to(unif, (g.shape[0],) + unif.shape).copy())

    return Workload(name=name, features=features, states_per_feature=states_per_feature,
                    component_names=component_names, groups=groups, concentration=concentration,
                    weights=weights, source=source, seeds=(data_seed, state_seed))


def algorithmic_bytes(n_objects, n_features, n_states, groups_per_component, n_patterns, packed=False):
    """SURVEY.md 8(d) contract figure B_eval: bytes one mixture-LL eval must touch.
    one-hot block (or N*F packed state ids when `packed`) + fp32 prob tables + fp32 weight
    tables + per-object group / pattern ids + the scalar result."""
    n_comp = len(groups_per_component)
    obs = n_objects * n_features * (1 if packed else n_states)
    tables = sum(groups_per_component) * n_features * n_states * 4
    w_tables = n_patterns * n_features * n_comp * 4
    ids = n_objects * (n_comp + 1)
    return obs + tables + w_tables + ids + 8


def unique_bytes_per_launch(n_objects, n_features, n_states, groups_per_component, n_patterns, n_evals, packed=False):
    """What a launch of `n_evals` resident states must touch when the feature block the states SHARE is counted once:
    the block + per state its tables, ids and result (algorithmic_bytes counts the block once per eval: the contract
    figure of SURVEY.md 8(d); this is the same inventory with the shared part not multiplied)."""
    obs = n_objects * n_features * (1 if packed else n_states)
    per_state = algorithmic_bytes(n_objects, n_features, n_states, groups_per_component, n_patterns, packed) - obs
    return obs + n_evals * per_state

