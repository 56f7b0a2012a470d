"""Boundary types: sbayes_amd/state.py must answer exactly like the
reference's Sample / GroupedParameters / CacheNode on a scripted edit sequence recorded from
the reference (tests/golden/state_versions.json, made by make_golden.state_fixture), plus the
reference's own test_state.py cases."""
import json
from collections import OrderedDict

import numpy as np

from sbayes_amd.state import (CacheNode, Confounder, FeatureCounts, GroupedParameters, ModelShapes, Sample)
from sbayes_amd.synthetic import make_workload
from tests._fixtures import GOLDEN


def mirror_sample():
    wl = make_workload("cfg1")
    n, f, s = wl.shape
    confounders = OrderedDict((name, Confounder(name, g)) for name, g in zip(wl.component_names[1:], wl.groups[1:]))
    shapes = ModelShapes(wl.clusters.shape[0], n, f, s, wl.states_per_feature, len(confounders),
                         {k: c.n_groups for k, c in confounders.items()})
    counts = {name: np.zeros((g.shape[0], f, s), dtype=np.float32) for name, g in zip(wl.component_names, wl.groups)}
    sample = Sample.from_numpy_arrays(wl.clusters.copy(), wl.weights.copy(), confounders, wl.source.copy(), counts, shapes)
    # the golden script ran after recalculate_feature_counts (one set_value per component)
    for name in wl.component_names:
        sample.feature_counts[name].set_value(sample.feature_counts[name].value.copy())
    return sample


def state_script(sample):
    """Same sequence as tests/golden/make_golden.py::state_script."""
    log = []

    def snap(tag, s):
        cl, cc = s.cache.component_likelihoods, s.cache.group_likelihoods["clusters"]
        log.append(dict(
            tag=tag,
            clusters_version=int(s.clusters.version), clusters_gv=[float(v) for v in s.clusters.group_versions],
            counts_version=int(s.feature_counts["clusters"].version),
            counts_gv=[float(v) for v in s.feature_counts["clusters"].group_versions],
            weights_version=int(s.weights.version), source_version=int(s.source.version),
            lh_outdated=bool(cl.is_outdated()),
            lh_changed=[int(v) for v in cl.what_changed(["clusters", "clusters_counts"], caching=True)],
            lh_changed_nocache=[int(v) for v in cl.what_changed(["clusters", "clusters_counts"], caching=False)],
            grp_changed=[int(v) for v in cc.what_changed("counts", caching=True)],
            w_outdated=bool(s.cache.weights_normalized.is_outdated()),
            has_components_col0=[bool(v) for v in s.cache.has_components.value[:, 0]],
            clusters_shared=bool(s.clusters.shared),
        ))

    snap("initial", sample)
    with sample.cache.component_likelihoods.edit():
        pass
    with sample.cache.group_likelihoods["clusters"].edit():
        pass
    sample.cache.weights_normalized.update_value(sample.cache.weights_normalized.value)
    snap("caches_up_to_date", sample)
    free = int(np.flatnonzero(~sample.clusters.value.any(axis=0))[0])
    sample.clusters.add_object(1, free)
    snap("add_object_cluster1", sample)
    cand = sample.copy()
    snap("after_copy_original", sample)
    snap("after_copy_candidate", cand)
    member = int(np.flatnonzero(cand.clusters.value[0])[0])
    cand.clusters.remove_object(0, member)
    snap("candidate_remove_object_cluster0", cand)
    snap("original_after_candidate_edit", sample)
    diff = np.zeros(cand.feature_counts["clusters"].value.shape, dtype=np.float32)
    diff[1, 2, 0] = 1.0
    cand.feature_counts["clusters"].add_changes(diff)
    snap("candidate_counts_add_changes_group1", cand)
    with cand.cache.component_likelihoods.edit():
        pass
    snap("candidate_lh_cache_refreshed", cand)
    cand.weights.set_value(cand.weights.value.copy())
    snap("candidate_weights_set_value", cand)
    with cand.source.edit() as src:
        src[0, 0, :] = False
    snap("candidate_source_edit", cand)
    cand.feature_counts["clusters"].set_value(cand.feature_counts["clusters"].value.copy())
    snap("candidate_counts_set_value", cand)
    cand.clusters.set_items((0, member), True)
    snap("candidate_clusters_set_items", cand)
    cand.everything_changed()
    snap("candidate_everything_changed", cand)
    return log


def test_mirror_matches_reference_version_trace():
    with open(GOLDEN / "state_versions.json") as fh:
        want = json.load(fh)
    got = state_script(mirror_sample())
    assert [g["tag"] for g in got] == [w["tag"] for w in want]
    for g, w in zip(got, want):
        assert g == w, g["tag"]


class TestGroupedParameters:
    """The reference's test/test_state.py cases."""

    def setup_method(self):
        self.param = GroupedParameters(np.arange(12).reshape((3, 4)))
        self.calc = CacheNode(np.empty((3, 4)))

    def test_initial_state(self):
        assert self.param.value[1, 2] == 6 and self.param.version == 0

    def test_set_items(self):
        self.param.set_items((1, 2), 1000)
        assert self.param.value[1, 2] == 1000 and self.param.version == 1
        assert self.param.group_versions.tolist() == [0, 1, 0]

    def test_edit(self):
        with self.param.edit() as value:
            value[1, 2] = 1000
        assert self.param.value[1, 2] == 1000 and self.param.version == 1
        assert self.param.group_versions.tolist() == [1, 1, 1]

    def test_set_value(self):
        new_value = self.param.value.copy()
        new_value[1, 2] = 1000
        self.param.set_value(new_value)
        assert self.param.value[1, 2] == 1000 and self.param.version == 1

    def test_read_only_between_edits_and_copy_on_write(self):
        assert not self.param.value.flags.writeable
        other = self.param.copy()
        assert other.value is self.param.value and other.shared and self.param.shared
        other.set_group(2, 0)
        assert other.value is not self.param.value
        assert self.param.value[2].tolist() == [8, 9, 10, 11] and other.value[2].tolist() == [0, 0, 0, 0]
        assert self.param.version == 0 and other.version == 1

    def test_cache_node_tracks_group_changes(self):
        self.calc.add_input("p", self.param)
        assert self.calc.is_outdated()
        assert self.calc.what_changed("p").tolist() == [0, 1, 2]
        self.calc.set_up_to_date()
        assert not self.calc.is_outdated() and self.calc.what_changed("p").tolist() == []
        self.param.set_group(1, 5)
        assert self.calc.is_outdated() and self.calc.what_changed("p").tolist() == [1]
        assert self.calc.what_changed("p", caching=False).tolist() == [0, 1, 2]
        assert self.calc.ahead_of("p")


def test_feature_counts_add_changes_marks_only_changed_groups():
    fc = FeatureCounts(np.zeros((3, 2, 2), dtype=np.float32))
    diff = np.zeros((3, 2, 2), dtype=np.float32)
    diff[2, 1, 0] = 1
    fc.add_changes(diff)
    assert fc.version == 1 and fc.group_versions.tolist() == [0, 0, 1]
    assert fc.value[2, 1, 0] == 1
