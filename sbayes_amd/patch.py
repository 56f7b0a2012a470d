"""install(): route real sBayes' likelihood path through the MI355X engine.

Patches the names through which the reference reaches the path:
  sbayes.model.likelihood.{Likelihood, compute_component_likelihood, update_weights, normalize_weights}
  sbayes.model.{Likelihood, ...} re-exports, sbayes.model.model.Likelihood (model.py:7, :45)
  sbayes.sampling.conditionals.{compute_component_likelihood, likelihood_per_component,
                                likelihood_per_component_subset, update_weights}   (conditionals.py:14)
  sbayes.sampling.counts.{compute_effect_counts, recalculate_feature_counts, update_feature_counts}
  sbayes.model.prior.{update_weights, normalize_weights}                          (prior.py:20)
Only when sBayes is importable; raises otherwise.  uninstall() restores the originals.

install(operators=True) additionally replaces the two heaviest consumers of the `[N, F, C]` component-likelihood
array inside the reference's operators by their device forms, so that array no longer
has to cross PCIe for them:
  sbayes.sampling.operators.AlterCluster.compute_cluster_posterior        (operators.py:1035-1073; inherited by
                                                                            AlterClusterWide)
  sbayes.sampling.operators.AlterClusterWide.compute_raw_cluster_probs    (operators.py:1420-1472)
  sbayes.sampling.operators.GibbsSampleSource.calculate_source_posterior  (operators.py:554-574)
  sbayes.sampling.operators.component_likelihood_given_unchanged          (operators.py:863-928)
  sbayes.sampling.loggers.LikelihoodLogger._write_sample                  (loggers.py:354-359)
  sbayes.model.prior.SourcePrior.__call__                                 (prior.py:573-611; per-object cache protocol kept)
  sbayes.sampling.operators.ClusterJump.get_jump_lh                       (operators.py:1679-1722, with
                                                                            expected_confounder_features :1342-1379)
  sbayes.sampling.operators.GibbsSampleWeights.source_lh_by_feature       (operators.py:677-685; _propose :597-636 runs
        UNCHANGED: update_weights hands it a lazily materialised array that knows its sample, and this static method
        evaluates that sample's resident state on the device -- the two [N, F, C] arrays per call are never built)
Proposal logic, RNG use and everything else of the operators stay the reference's.

Process model (sbayes_amd/_proc.py, INTEGRATION.md "Processes").  The reference starts MC3 workers and run pools with
`multiprocessing`'s default start method -- fork on Linux (mcmc_setup.py:271-282, cli.py:104-109) -- and a HIP context does
not survive fork().  install(mp_start_method="forkserver" | "spawn") fixes the start method before the first worker exists;
with either, a worker is a fresh interpreter in which nothing is patched yet, so a pickled `Likelihood` re-installs the
patch (same `operators` choice) when it is unpickled there -- the model is the first thing a worker receives
(`send_initialize_chain`, mcmc_setup.py:299, :554), before it builds its MCMCChain and operators."""
from __future__ import annotations

import hashlib
import importlib
import os
import inspect
import warnings

_SAVED = []
_ADDED = []              # (class, attribute) pairs install() ADDED to reference classes (removed by uninstall())
_INSTALLED = None        # {"operators": bool} while the patch is installed in this process

# A replaced method shadows whatever the reference does there, so each device form is tied to the reference body it
# mirrors: SHA-1 of the method's source text with whitespace runs collapsed, taken from the reference revision this
# package was written against.  install(operators=True) warns when an installed sBayes differs (nit 9):
# the swap still happens, but not silently.
MIRRORED_SOURCES = {
    "ClusterJump.get_jump_lh": "dcb154279abc275f24d212722ed22fdf83e18041",
    "GibbsSampleWeights._propose": "c68637810ba27a9cd84c54087ae1341861046e9a",
    "GibbsSampleWeights.source_lh_by_feature": "0952193be39ab7f1fb99c1fe64fa88568e3d9312",
    "AlterCluster.compute_cluster_posterior": "380025299a64acc90921c0fe29fd9fcc70c8410a",
    "AlterClusterWide.compute_raw_cluster_probs": "71b78b6ef2cd63cd72091c9f90d787df26b9fa97",
    "AlterCluster.compute_feature_weights_with_and_without": "0cf7022db29a8d50a4395338c604350fac790eb6",
    "GibbsSampleSource.calculate_source_posterior": "042ce6497bc2b08ffe197946aeaeab98a5e24bce",
    "sbayes.sampling.operators.component_likelihood_given_unchanged": "de5b027624d8597d43f08e2577d0035607c22d51",
    "LikelihoodLogger._write_sample": "5f892d865e576862f5e1730acbf3b18efc7af9da",
    "ClusterEffectProposals.expected_confounder_features": "efb685c5e0b1f2818dde1f2244d6f0ff086107cc",
    "SourcePrior.__call__": "c76b2825409bc280761a6b598113f74c9b395dbb",
    "GibbsSampleSource._propose": "61911f12df7948eea0f207e45f49d97d7e0e5352",
    "ClusterOperator.gibbs_sample_source": "4c1f20c2821f64d9eed7ad4e8fa9cf50a76d0de1",
    "FeatureCounts.add_changes": "7c38d633f8488c553f1a031f57751a7db58deed2",
    # the cache-node protocol the host layer runs in native code for nodes of exactly this class (_register_cache_nodes)
    "CacheNode.is_outdated": "7acdcaadcdb930b67a958d4e92cb6a22fbd99f68",
    "CacheNode.ahead_of": "5cc191e87c93f012e12e18dda747d0d29c5f63de",
    "CacheNode.what_changed": "897c8ba27633cf13b1a96dee47fba5a662c43cc7",
    "CacheNode.set_up_to_date": "3d3d7d5348f3e7d579936415b3c978a88b910a54",
    "CacheNode.update_value": "207072adb3b1f5857f9751156541e3b11d2fa7cf",
    "CacheNode.edit": "f97842f4ecfda2357d10973488fedec3bf0f4415",
    "CacheNode.version": "9f3783fe9344800ec2dd6711a391f439d17a9020",
    "CacheNode.value": "2c599b691219e9e677be97749fab9ee166bf0018",
    # read-only properties the native bind reads from the instance (_register_trusted_classes)
    "Sample.clusters": "1c35d1fc15ec7e57e6f30149f4f21bdd0eebe6c7",
    "Sample.weights": "f122fc33db30aa3a14db2a268452a137ad71f5cc",
    "Sample.source": "a40fb15cd7c42bd6982c31b029f11cd455c1c9d3",
    "Sample.feature_counts": "086f7821a38fd065ba46bd69d7a710c18beeeb26",
    "Parameter.value": "2c599b691219e9e677be97749fab9ee166bf0018",
    "ConfoundingEffectsPrior.concentration_array": "a089466a31285e330baf46ecc81f435fcd63a359",
    "ArrayParameter.resolve_sharing": "9456c9d76c849a1e341c66b73e69d630c52fb3bf",
    "GroupedParameters.resolve_sharing": "e26fbdc96e79af850e2840a33297ddb0d997d967",
}


def source_digest(obj) -> str:
    """SHA-1 of an object's source text, whitespace runs collapsed (formatting changes do not count)."""
    return hashlib.sha1(" ".join(inspect.getsource(obj).split()).encode()).hexdigest()


def _check_mirrored(owner, name):
    key = f"{getattr(owner, '__name__', owner)}.{name}"
    want = MIRRORED_SOURCES.get(key)
    try:
        got = source_digest(inspect.getattr_static(owner, name) if inspect.isclass(owner) else getattr(owner, name))
    except (OSError, TypeError, AttributeError):
        return
    if want is not None and got != want:
        warnings.warn(f"sbayes_amd.patch: the installed sBayes' {key} differs from the revision its device form mirrors; "
                      f"the device form replaces it anyway -- re-check sbayes_amd/operators.py against it", RuntimeWarning)


def installed():
    """{"operators": bool} if install() ran in this process (and uninstall() has not), else None."""
    return dict(_INSTALLED) if _INSTALLED is not None else None


def set_mp_start_method(method):
    """Fix multiprocessing's start method to one that does not copy a HIP context into workers ("forkserver" or
    "spawn"; "auto" = "forkserver" unless the application already fixed one).  Raises if the application fixed "fork"
    and asks for something else here -- that is its decision to revisit, not ours to override."""
    import multiprocessing as mp
    current = mp.get_start_method(allow_none=True)
    if method == "auto":
        if current is None:
            mp.set_start_method("forkserver")
        elif current == "fork":
            warnings.warn("sbayes_amd.patch: multiprocessing's start method is fixed to 'fork'; workers forked after this "
                          "process first used the GPU cannot use it (sbayes_amd._proc.ForkedWithHipError).  Use "
                          "'forkserver' or 'spawn'.", RuntimeWarning, stacklevel=3)
        return mp.get_start_method(allow_none=True)
    if method not in ("forkserver", "spawn"):
        raise ValueError(f"mp_start_method must be 'forkserver', 'spawn' or 'auto', got {method!r}")
    if current is None:
        mp.set_start_method(method)
    elif current != method:
        raise RuntimeError(f"sbayes_amd.patch.install(mp_start_method={method!r}): multiprocessing's start method is "
                           f"already fixed to {current!r}")
    return method


def install(operators=False, mp_start_method=None, gibbs_source=False):
    """operators=True: the device forms listed in the module docstring.  gibbs_source=True (implies operators): the two
    Gibbs source resamplings of the reference run on the device, with the uniforms np.random yields
    at the point where the reference's `sample_categorical` draws them (draw for draw: the same Markov chain) --
      GibbsSampleSource._propose (operators.py:495-552), its body after `select_object_subset`: posterior, draw, new source
        rows, count delta, both transition log-probabilities in ONE engine call (operators.gibbs_sample_source ->
        sbe_gibbs_propose);
      ClusterOperator.gibbs_sample_source (operators.py:796-851), the source resampling inside every AlterCluster /
        AlterClusterWide / ClusterJump proposal: likelihood under the kept observations, both posteriors, draw, selected
        probabilities AND the count delta the reference asks for next (update_feature_counts, :827) in ONE engine call
        (operators.cluster_gibbs_sample_source -> sbe_given_unchanged_gibbs_counts).
    At the headline shape these two bodies are the largest items of the reference's per-step Python."""
    global _INSTALLED
    operators = bool(operators) or bool(gibbs_source)
    if mp_start_method is not None:
        set_mp_start_method(mp_start_method)
    from . import conditionals as my_cond
    from . import counts as my_counts
    from . import likelihood as my_lik
    try:
        lik = importlib.import_module("sbayes.model.likelihood")
        model_pkg = importlib.import_module("sbayes.model")
        model_mod = importlib.import_module("sbayes.model.model")
        cond = importlib.import_module("sbayes.sampling.conditionals")
        counts = importlib.import_module("sbayes.sampling.counts")
    except ImportError as exc:
        raise RuntimeError("sbayes_amd.patch.install(): sBayes is not importable") from exc
    # import every module that binds the names BEFORE swapping anything: a module first imported after
    # the swap would bind the replacements as its "originals" and uninstall() could not restore them
    importers = []
    for modname in ("sbayes.sampling.operators", "sbayes.sampling.initializers", "sbayes.sampling.loggers",
                    "sbayes.sampling.mcmc", "sbayes.sampling.mcmc_chain", "sbayes.mcmc_setup", "sbayes.model.prior"):
        try:
            importers.append(importlib.import_module(modname))
        except ImportError:
            continue

    def swap(mod, name, new):
        if hasattr(mod, name) and getattr(mod, name) is not new:
            # (a class attribute is saved as it sits in the class dict: a staticmethod must come back as one)
            old = mod.__dict__[name] if inspect.isclass(mod) and name in mod.__dict__ else getattr(mod, name)
            _SAVED.append((mod, name, old))
            setattr(mod, name, new)

    for mod in (lik, model_pkg, model_mod):
        swap(mod, "Likelihood", my_lik.Likelihood)
    for mod in (lik, model_pkg, cond):
        swap(mod, "compute_component_likelihood", my_lik.compute_component_likelihood)
        swap(mod, "update_weights", my_lik.update_weights)
        swap(mod, "normalize_weights", my_lik.normalize_weights)
    swap(cond, "likelihood_per_component", my_cond.likelihood_per_component)
    swap(cond, "likelihood_per_component_subset", my_cond.likelihood_per_component_subset)
    for mod in (counts, lik, cond):
        swap(mod, "compute_effect_counts", my_counts.compute_effect_counts)
        swap(mod, "recalculate_feature_counts", my_counts.recalculate_feature_counts)
        swap(mod, "update_feature_counts", my_counts.update_feature_counts)
    _install_sparse_add_changes()
    _register_cache_nodes()
    _register_trusted_classes()
    if operators:
        _install_operator_forms(swap)
    if gibbs_source:
        _install_gibbs_source_form(swap)
    for m in importers:
        for name, new in (("likelihood_per_component", my_cond.likelihood_per_component),
                          ("update_weights", my_lik.update_weights),
                          ("normalize_weights", my_lik.normalize_weights),
                          ("recalculate_feature_counts", my_counts.recalculate_feature_counts),
                          ("update_feature_counts", my_counts.update_feature_counts),
                          ("compute_effect_counts", my_counts.compute_effect_counts),
                          ("compute_component_likelihood", my_lik.compute_component_likelihood)):
            swap(m, name, new)
    _INSTALLED = {"operators": bool(operators) or bool(_INSTALLED and _INSTALLED["operators"]),
                  "gibbs_source": bool(gibbs_source) or bool(_INSTALLED and _INSTALLED.get("gibbs_source"))}


_NODE_CLASSES = []


def _register_cache_nodes():
    """The reference's CacheNode protocol (sbayes/sampling/state.py:215-321: is_outdated, ahead_of, what_changed, set_up_to_date,
    edit, version, value) runs in native code inside the host layer's own functions (csrc/sbe_pyhost.c: node_*, likelihood_call,
    store_per_object) for nodes of EXACTLY the reference's CacheNode class -- and only when every one of those methods is the
    revision the native code mirrors; otherwise (and for subclasses such as HasComponents) the node's own methods are called."""
    from . import _fast
    try:
        state = importlib.import_module("sbayes.sampling.state")
        cls, grouped = state.CacheNode, state.GroupedParameters
        for name in ("is_outdated", "ahead_of", "what_changed", "set_up_to_date", "update_value", "edit", "version", "value"):
            obj = inspect.getattr_static(cls, name)
            if isinstance(obj, property):
                obj = obj.fget
            if source_digest(obj) != MIRRORED_SOURCES[f"CacheNode.{name}"]:
                return
    except (ImportError, AttributeError, OSError, TypeError):
        return
    _fast.register_node_classes(cls, grouped)
    _NODE_CLASSES.append((cls, grouped))


_TRUSTED_CLASSES = []


def _register_trusted_classes():
    """The reference's Sample.clusters / .weights / .source / .feature_counts and Parameter.value are properties that return the
    attribute of the same name with a leading underscore (sbayes/sampling/state.py:30-32, 578-592), and a static
    ConfoundingEffectsPrior.concentration_array(sample) returns self._concentration_array (sbayes/model/prior.py:325-354).  The
    native bind (csrc/sbe_pyhost.c) reads those attributes from the instance for objects of EXACTLY these classes -- registered
    only when the source of each property / method is the revision mirrored, and no parameter subclass overrides `value`."""
    from . import _fast

    def same(owner, name, key):
        obj = inspect.getattr_static(owner, name)
        if isinstance(obj, property):
            obj = obj.fget
        return source_digest(obj) == MIRRORED_SOURCES[key]

    samples, params, conf_priors, count_classes = [], [], [], []
    try:
        state = importlib.import_module("sbayes.sampling.state")
        # FeatureCounts' copy-on-write step (add_changes -> resolve_sharing) in the native add_rows_many: the inherited two-liner
        if (same(state.ArrayParameter, "resolve_sharing", "ArrayParameter.resolve_sharing")
                and same(state.GroupedParameters, "resolve_sharing", "GroupedParameters.resolve_sharing")
                and inspect.getattr_static(state.FeatureCounts, "resolve_sharing") is inspect.getattr_static(state.GroupedParameters, "resolve_sharing")):
            count_classes.append(state.FeatureCounts)
        if all(same(state.Sample, n, f"Sample.{n}") for n in ("clusters", "weights", "source", "feature_counts")):
            samples.append(state.Sample)
        if same(state.Parameter, "value", "Parameter.value"):
            base = inspect.getattr_static(state.Parameter, "value")
            for name in ("ArrayParameter", "GroupedParameters", "Clusters", "FeatureCounts"):
                cls = getattr(state, name, None)
                if inspect.isclass(cls) and inspect.getattr_static(cls, "value") is base:
                    params.append(cls)
    except (ImportError, AttributeError, OSError, TypeError):
        pass
    try:
        prior = importlib.import_module("sbayes.model.prior")
        if same(prior.ConfoundingEffectsPrior, "concentration_array", "ConfoundingEffectsPrior.concentration_array"):
            conf_priors.append(prior.ConfoundingEffectsPrior)
    except (ImportError, AttributeError, OSError, TypeError):
        pass
    _fast.register_trusted(samples, params, conf_priors, count_classes)
    _TRUSTED_CLASSES.append((samples, params, conf_priors, count_classes))


def _install_sparse_add_changes():
    """FeatureCounts.add_changes_rows(group_idx, rows): add_changes (sbayes/sampling/state.py:340-350) for a difference
    that is zero outside the listed rows -- what the drop-in update_feature_counts has in hand.  Same value, version and
    group versions as add_changes(dense diff); the [n_groups, F, S] zero array, the whole-table add and the whole-table
    compare per component and call are not built (a third of update_feature_counts' host time, tools/host_residual.py).
    Added only when the installed add_changes is the body this mirrors; otherwise the dense call stays."""
    try:
        state = importlib.import_module("sbayes.sampling.state")
        cls = state.FeatureCounts
        if source_digest(inspect.getattr_static(cls, "add_changes")) != MIRRORED_SOURCES["FeatureCounts.add_changes"]:
            return
    except (ImportError, AttributeError, OSError, TypeError):
        return
    if "add_changes_rows" in cls.__dict__:
        return
    import numpy as np

    def add_changes_rows(self, group_idx, rows):
        if self.shared:
            self.resolve_sharing()
        if len(group_idx):
            self._value.flags.writeable = True
            self._value[group_idx] += rows
            self._value.flags.writeable = False
        self.version += 1
        if len(group_idx):
            self.group_versions[group_idx[np.any(rows != 0, axis=(1, 2))]] = self.version

    cls.add_changes_rows = add_changes_rows
    _ADDED.append((cls, "add_changes_rows"))


def _install_gibbs_source_form(swap):
    from . import operators as my_ops
    ref_ops = importlib.import_module("sbayes.sampling.operators")
    _check_mirrored(ref_ops.GibbsSampleSource, "_propose")

    def _propose(self, sample, object_subset=slice(None), **kwargs):
        """GibbsSampleSource._propose (operators.py:495-552): the subset comes from the reference's own
        select_object_subset (its RNG use unchanged); everything after it runs on the device."""
        object_subset = self.select_object_subset(sample)
        return my_ops.gibbs_sample_source(self.model, sample, object_subset, self.temperature, self.prior_temperature,
                                          self.sample_from_prior)

    swap(ref_ops.GibbsSampleSource, "_propose", _propose)

    # ClusterOperator.gibbs_sample_source (operators.py:796-851): the source resampling of the cluster proposals
    _check_mirrored(ref_ops.ClusterOperator, "gibbs_sample_source")
    reference_cluster_gibbs = ref_ops.ClusterOperator.__dict__["gibbs_sample_source"]

    def gibbs_sample_source(self, sample_new, sample_old, i_cluster, object_subset=slice(None)):
        out = my_ops.cluster_gibbs_sample_source(self.model, sample_new, sample_old, i_cluster, object_subset, self.temperature,
                                                 self.prior_temperature, self.sample_from_prior)
        if out is None:                                          # dynamic priors: the reference's own body
            return reference_cluster_gibbs(self, sample_new, sample_old, i_cluster, object_subset=object_subset)
        return out

    swap(ref_ops.ClusterOperator, "gibbs_sample_source", gibbs_sample_source)


def _install_operator_forms(swap):
    import numpy as np

    from . import likelihood as my_lik
    from . import operators as my_ops
    ref_ops = importlib.import_module("sbayes.sampling.operators")
    # with the device forms below nothing in the sampling loop computes cache.component_likelihoods any more: a sample's stale
    # 8 N F C byte block stops being copied by every Sample.copy() (likelihood.LazyBlock; Likelihood.__call__ swaps it in)
    if os.environ.get("SBAYES_AMD_LEAN_SAMPLES", "1") != "0":
        swap(my_lik, "LEAN_SAMPLES", True)

    def compute_cluster_posterior(self, sample, i_cluster, available):
        """AlterCluster.compute_cluster_posterior (operators.py:1035-1073) on the device."""
        if self.sample_from_prior or not self.gibbsish:
            return 0.5 * np.ones(np.count_nonzero(available))
        geo = None
        if self.consider_geo_prior:
            geo = np.exp(self.model.prior.geo_prior.get_costs_per_object(sample, i_cluster)[available]
                         / self.prior_temperature)
        return my_ops.compute_cluster_posterior(self.model, sample, i_cluster, available, self.temperature,
                                                self.prior_temperature, self.additive_smoothing, geo)

    def calculate_source_posterior(self, sample, object_subset=slice(None)):
        """GibbsSampleSource.calculate_source_posterior (operators.py:554-574) on the device."""
        return my_ops.calculate_source_posterior(self.model, sample, object_subset, self.temperature,
                                                 self.prior_temperature)

    def compute_raw_cluster_probs(self, sample, i_cluster, available):
        """AlterClusterWide.compute_raw_cluster_probs (operators.py:1420-1472): the candidate table comes from the
        reference's own cluster_effect_proposal, the marginals from the device; the geo-prior part is the reference's."""
        model = self.model
        if self.sample_from_prior:
            return 0.5 * np.ones(np.count_nonzero(available))
        p = self.cluster_effect_proposal(model, sample, i_cluster, self.temperature, self.prior_temperature)
        with np.errstate(under="ignore"):
            marginal_lh_z01 = np.exp(my_ops.cluster_log_marginals(model, sample, p, available, self.temperature,
                                                                  self.prior_temperature))
        if self.consider_geo_prior:
            if self.cluster_effect_proposal is ref_ops.ClusterEffectProposals.residual_counts:
                distances = model.data.geo_cost_matrix[available][:, available]
                z = ref_ops.normalize(marginal_lh_z01[1] / (marginal_lh_z01[0] + marginal_lh_z01[1] + ref_ops.EPS))
                log_geo_prior_ratio = -z.dot(distances) / model.prior.geo_prior.scale
            else:
                log_geo_prior_ratio = model.prior.geo_prior.get_costs_per_object(sample, i_cluster)[available]
            marginal_lh_z01[1] *= np.exp(log_geo_prior_ratio / self.prior_temperature / self.geo_scaler)
        return marginal_lh_z01[1] / (marginal_lh_z01[0] + marginal_lh_z01[1] + ref_ops.EPS)

    def get_jump_lh(self, sample, i_source_cluster, i_target_cluster):
        """ClusterJump.get_jump_lh (operators.py:1679-1722) on the device."""
        return my_ops.jump_lh(self.model, sample, i_source_cluster, i_target_cluster, self.temperature,
                              self.prior_temperature)

    reference_source_lh = ref_ops.GibbsSampleWeights.__dict__["source_lh_by_feature"].__func__

    def source_lh_by_feature(source, weights, na_features):
        """GibbsSampleWeights.source_lh_by_feature (operators.py:677-685).  The reference's _propose (:597-636) runs
        UNCHANGED and hands over `update_weights(sample)` -- here a lazily materialised NormalizedWeights that knows its
        sample: when `source` and the weights are that sample's current ones, the per-feature sums come from the
        sample's RESIDENT state on the device (changed source rows, group ids and F*C weights go up, F floats come
        back; the [N, F, C] array is never built).  The engine is the one whose resident NA mask equals `na_features`
        (registry.engine_for_observations).  Any other argument combination -- and a mask no live engine holds -- is
        served by the reference's own expression on the materialised array.  Installed once, no name is swapped while a proposal runs."""
        from .binding import _bind_slot
        from .likelihood import NormalizedWeights
        from . import registry
        sample = weights.sample_if_current() if isinstance(weights, NormalizedWeights) else None
        if sample is not None and source is sample.source.value:
            # the engine is identified by the NA mask the caller hands over (it decides which observations count):
            # two datasets of one (N, F, C) shape in one process never share an engine here
            # (only the layout computation and the engine lookup may decline -- a sample object of another form, a mask no
            #  live engine holds: the reference expression below; an error of the bind or of the engine call is a real error
            #  and propagates)
            try:
                layout = [sample.clusters.value.shape[0]] + [c.group_assignment.shape[0] for c in sample.confounders.values()]
                eng = registry.engine_for_observations(na_features, np.shape(source)[2], layout)
            except (ValueError, AttributeError):
                eng = None
            if eng is not None:
                _bind_slot(eng, None, sample, 0, with_source=True)
                return eng.source_lh_by_feature(0)
        return reference_source_lh(source, np.asarray(weights), na_features)

    # SourcePrior.__call__ (prior.py:573-611): per-object log prior from the device, the reference's cache protocol kept
    try:
        ref_prior = importlib.import_module("sbayes.model.prior")
    except ImportError:
        ref_prior = None
    if ref_prior is not None and hasattr(ref_prior, "SourcePrior"):
        from . import conditionals as my_cond_sp
        _check_mirrored(ref_prior.SourcePrior, "__call__")
        reference_source_prior = ref_prior.SourcePrior.__call__

        def source_prior_call(self, sample, caching=True):
            owner = getattr(self, "_sbayes_amd_owner", None)     # left by sbayes_amd.likelihood.Likelihood.__init__
            if owner is None:                                      # a SourcePrior outside a (patched) Model
                return reference_source_prior(self, sample, caching=caching)
            return my_cond_sp.source_prior(owner, sample, caching=caching)

        swap(ref_prior.SourcePrior, "__call__", source_prior_call)
        swap(my_cond_sp, "DEVICE_SOURCE_PRIOR", True)            # (Likelihood.__call__ takes the source prior along)


    for owner, name in ((ref_ops.ClusterJump, "get_jump_lh"), (ref_ops.GibbsSampleWeights, "_propose"),
                        (ref_ops.GibbsSampleWeights, "source_lh_by_feature"),
                        (ref_ops.AlterCluster, "compute_cluster_posterior"),
                        (ref_ops.AlterClusterWide, "compute_raw_cluster_probs"),
                        (ref_ops.AlterCluster, "compute_feature_weights_with_and_without"),
                        (ref_ops.GibbsSampleSource, "calculate_source_posterior"),
                        (ref_ops, "component_likelihood_given_unchanged"),
                        (ref_ops.ClusterEffectProposals, "expected_confounder_features")):
        if hasattr(owner, name):
            _check_mirrored(owner, name)
    swap(ref_ops.ClusterJump, "get_jump_lh", get_jump_lh)
    swap(ref_ops.GibbsSampleWeights, "source_lh_by_feature", staticmethod(source_lh_by_feature))
    swap(ref_ops.AlterCluster, "compute_cluster_posterior", compute_cluster_posterior)
    swap(ref_ops.AlterClusterWide, "compute_raw_cluster_probs", compute_raw_cluster_probs)
    swap(ref_ops.GibbsSampleSource, "calculate_source_posterior", calculate_source_posterior)
    # module-level function with the reference's own signature (operators.py:863-928)
    swap(ref_ops, "component_likelihood_given_unchanged", my_ops.component_likelihood_given_unchanged)
    # the per-observation likelihood row of the LikelihoodLogger (loggers.py:354-359): one [N, F] float64 array from
    # the device instead of the [N, F, C] leave-one-out array and the [N, F, C] weights
    try:
        ref_loggers = importlib.import_module("sbayes.sampling.loggers")
    except ImportError:
        ref_loggers = None
    if ref_loggers is not None and hasattr(ref_loggers, "LikelihoodLogger"):
        from . import conditionals as my_cond
        _check_mirrored(ref_loggers.LikelihoodLogger, "_write_sample")

        def _write_sample(self, sample):
            lh = my_cond.observation_likelihoods(self.model, sample, exact=True).ravel()
            self.logged_likelihood_array.append(lh[None, ...])
            self.file.flush()

        swap(ref_loggers.LikelihoodLogger, "_write_sample", _write_sample)


def uninstall():
    global _INSTALLED
    _INSTALLED = None
    while _SAVED:
        mod, name, old = _SAVED.pop()
        setattr(mod, name, old)
    while _ADDED:
        cls, name = _ADDED.pop()
        if name in cls.__dict__:
            delattr(cls, name)
    while _NODE_CLASSES:
        from . import _fast
        _fast.unregister_node_classes(*_NODE_CLASSES.pop())
    while _TRUSTED_CLASSES:
        from . import _fast
        _fast.unregister_trusted(*_TRUSTED_CLASSES.pop())
