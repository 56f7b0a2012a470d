"""patch.install() swaps the engine-backed functions into a real sBayes by module-level name
(SURVEY.md 8(b)).  Needs the reference importable, so it runs only in the build container
(skipped on the GPU box, where /root/reference does not exist)."""
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference sBayes not present")


def test_install_and_uninstall_swap_names():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import _ref_stubs
    _ref_stubs.install()
    import sbayes.model.likelihood as ref_lik
    import sbayes.model.model as ref_model
    import sbayes.sampling.conditionals as ref_cond
    import sbayes.sampling.counts as ref_counts
    import sbayes.sampling.operators as ref_ops

    from sbayes_amd import conditionals, counts, likelihood, patch
    orig = (ref_model.Likelihood, ref_cond.compute_component_likelihood, ref_cond.likelihood_per_component,
            ref_counts.update_feature_counts, ref_ops.update_weights)
    patch.install()
    try:
        assert ref_model.Likelihood is likelihood.Likelihood
        assert ref_lik.Likelihood is likelihood.Likelihood
        assert ref_cond.compute_component_likelihood is likelihood.compute_component_likelihood
        assert ref_cond.likelihood_per_component is conditionals.likelihood_per_component
        assert ref_counts.update_feature_counts is counts.update_feature_counts
        assert ref_ops.update_weights is likelihood.update_weights
        assert ref_ops.likelihood_per_component is conditionals.likelihood_per_component
    finally:
        patch.uninstall()
    assert (ref_model.Likelihood, ref_cond.compute_component_likelihood, ref_cond.likelihood_per_component,
            ref_counts.update_feature_counts, ref_ops.update_weights) == orig


def test_operator_forms_install_and_uninstall():
    """install(operators=True): the prior module's names and SourcePrior.__call__ are swapped, the
    static method GibbsSampleWeights.source_lh_by_feature is replaced ONCE (no name is swapped while a proposal runs:
    ADVICE r2) and comes back as a staticmethod on uninstall."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import _ref_stubs
    _ref_stubs.install()
    import sbayes.model.prior as ref_prior
    import sbayes.sampling.operators as ref_ops

    from sbayes_amd import likelihood, patch
    orig_static = ref_ops.GibbsSampleWeights.__dict__["source_lh_by_feature"]
    orig_propose = ref_ops.GibbsSampleWeights.__dict__["_propose"]
    orig_sp = ref_prior.SourcePrior.__dict__["__call__"]
    orig_uw = ref_prior.update_weights
    assert isinstance(orig_static, staticmethod)
    patch.install(operators=True)
    try:
        assert ref_prior.update_weights is likelihood.update_weights
        assert ref_prior.normalize_weights is likelihood.normalize_weights
        assert ref_prior.SourcePrior.__dict__["__call__"] is not orig_sp
        assert isinstance(ref_ops.GibbsSampleWeights.__dict__["source_lh_by_feature"], staticmethod)
        assert ref_ops.GibbsSampleWeights.__dict__["source_lh_by_feature"] is not orig_static
        assert ref_ops.GibbsSampleWeights.__dict__["_propose"] is orig_propose          # the reference's own body runs
        assert "SourcePrior.__call__" in patch.MIRRORED_SOURCES
    finally:
        patch.uninstall()
    assert ref_ops.GibbsSampleWeights.__dict__["source_lh_by_feature"] is orig_static
    assert ref_prior.SourcePrior.__dict__["__call__"] is orig_sp and ref_prior.update_weights is orig_uw


def test_native_protocol_is_registered_only_for_the_mirrored_revision(monkeypatch):
    """patch.install() hands the reference's CacheNode / Sample / parameter / prior classes to the native host flow
    (csrc/sbe_pyhost.c: node_*, trust_setup) only while the source of every method it stands in for is the revision mirrored
    (patch.MIRRORED_SOURCES); a differing method leaves the class unregistered -- the nodes' own methods serve -- and uninstall()
    takes the registrations back."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import _ref_stubs
    _ref_stubs.install()
    import sbayes.model.prior as ref_prior
    import sbayes.sampling.state as ref_state

    from sbayes_amd import _fast, patch
    patch.install(operators=True)
    try:
        assert ref_state.CacheNode in _fast._NODE_PLAIN and ref_state.GroupedParameters in _fast._NODE_GROUPED
        samples, params, conf_priors, count_classes = _fast._TRUSTED
        assert ref_state.Sample in samples and ref_state.FeatureCounts in params and ref_state.Clusters in params
        assert ref_prior.ConfoundingEffectsPrior in conf_priors and ref_state.FeatureCounts in count_classes
        assert ref_state.HasComponents not in _fast._NODE_PLAIN                  # (overrides `value`: never by exact-class match)
    finally:
        patch.uninstall()
    assert ref_state.CacheNode not in _fast._NODE_PLAIN and ref_state.Sample not in _fast._TRUSTED[0]
    # another revision of one method: nothing of that family is registered
    monkeypatch.setitem(patch.MIRRORED_SOURCES, "CacheNode.what_changed", "0" * 40)
    monkeypatch.setitem(patch.MIRRORED_SOURCES, "Sample.source", "0" * 40)
    monkeypatch.setitem(patch.MIRRORED_SOURCES, "GroupedParameters.resolve_sharing", "0" * 40)
    patch.install(operators=True)
    try:
        assert ref_state.CacheNode not in _fast._NODE_PLAIN
        assert ref_state.Sample not in _fast._TRUSTED[0] and ref_state.FeatureCounts not in _fast._TRUSTED[3]
        assert ref_state.FeatureCounts in _fast._TRUSTED[1]                       # (Parameter.value is still the mirrored one)
    finally:
        patch.uninstall()
