"""`any_dynamic_priors = True` on the device: tests/_dynamic_prior_case.py against the reference's
recorded outputs -- collapsed likelihood across a hyperprior change, likelihood_per_component bit-exact, and the dynamic
branch of component_likelihood_given_unchanged (sbe_effect_counts + sbe_normalize_tables + sbe_subset_lh)."""
import pytest

from sbayes_amd.registry import release_all
from tests import _dynamic_prior_case as case

pytestmark = pytest.mark.gpu


def test_dynamic_prior_branches_against_the_reference():
    try:
        z, meta, wl = case.load()
        model, sample, new = case.drive(z, meta, wl)
        assert type(model.likelihood.engine).__module__ == "sbayes_amd.engine"
    finally:
        release_all()
