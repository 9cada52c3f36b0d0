"""A parity test whose PRECONDITION is not met (a fixture keyed to other bits, a storage format or kernel other than the one the test is
about) must be a RED test, not a skipped one: `pytest -x -q` stays green on a skip and the comparison silently stops being made
(VERDICT round 5, weak #4).  LPVS_ALLOW_STALE_FIXTURE=1 -- which the driver never sets -- turns it back into a skip for a developer
who is in the middle of regenerating a fixture."""
import os

import pytest


def precondition_not_met(msg: str):
    if os.environ.get("LPVS_ALLOW_STALE_FIXTURE") == "1":
        pytest.skip(msg)
    pytest.fail(msg + "  [LPVS_ALLOW_STALE_FIXTURE=1 skips instead]", pytrace=False)
