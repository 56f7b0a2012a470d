"""The REAL sampler's engine-level call sequence on the real device.

tests/golden/*_calls.npz were recorded in the build container: the reference's own sampler (initialiser, operators,
MH loop) running on the drop-in host layer under patch.install(operators=True), with the device replaced by a
recording double.  Here the same sequence -- same slot uploads in the same order, same partial re-binds, same
operator-form evaluations with the same argument arrays -- runs against the real Engine through the C ABI, and every
result is checked against the recorded one (bit-exact where the reference's arithmetic is reproduced bit for bit,
documented tolerances otherwise: tests/_call_log.py:COMPARE)."""
import pytest

from sbayes_amd.engine import Engine
from tests._call_log import replay
from tests._fixtures import GOLDEN
from tests.test_call_log_cpu import features_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["test_files", "south_america", "cfg1", "headline", "south_america_gibbs", "headline_gibbs", "cfg1_gibbs"])
def test_recorded_sampler_calls_on_the_device(tag):
    feats = features_of(tag)
    counts, meta = replay(GOLDEN / f"{tag}_calls.npz", lambda n_groups: Engine(feats, n_groups, n_slots=4))
    assert sum(counts.values()) > 400 and counts["__step__"] >= 48      # (fewer calls every round: 633 -> 491 for 48 headline steps)
