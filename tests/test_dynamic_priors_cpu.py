"""`any_dynamic_priors = True` on the oracle-backed double: the host logic of the drop-in layer
(hyperprior_has_changed, the content-compared bind of a concentration table that is rewritten in place, the
dynamic branch of component_likelihood_given_unchanged) against the REFERENCE's recorded outputs."""
import numpy as np

from sbayes_amd import binding, conditionals, counts, likelihood, registry
from tests import _dynamic_prior_case as case
from tests._fake_engine import make_get_engine


def test_dynamic_prior_branches_against_the_reference(monkeypatch):
    engines = {}
    get_engine = make_get_engine(engines)
    for mod in (registry, likelihood, conditionals, counts, binding):
        monkeypatch.setattr(mod, "get_engine", get_engine, raising=True)
    z, meta, wl = case.load()
    case.drive(z, meta, wl)
    eng = next(iter(engines.values()))
    kinds = [c[0] for c in eng.calls]
    assert "subset_lh" in kinds                     # the dynamic branch of component_likelihood_given_unchanged ran
    assert "given_unchanged_lh" not in kinds        # ... not the static-prior resident form


def test_mirror_prior_restates_the_reference_prior():
    z, meta, wl = case.load()
    model, sample, dynamic = case.build(z, meta, wl)
    for i, k in enumerate(meta["component_names"]):
        sample.feature_counts[k].set_value(z[f"s0_counts_{i}"])
    a = dynamic.concentration_array(sample)
    assert a.dtype == np.float64 and a.flags.writeable and np.array_equal(a, z["s0_conc_2"])
    assert dynamic.concentration_array(sample) is a            # the SAME array object, rewritten in place (prior.py:349-352)
