import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tmp_models(tmp_path_factory):
    """Directory of synthetic ggml model files, written once per session on demand."""
    from openhush_amd import modelfile, synth
    d = tmp_path_factory.mktemp("models")
    cache = {}

    def get(preset: str, seed: int = 1234) -> str:
        key = (preset, seed)
        if key not in cache:
            path = os.path.join(str(d), f"ggml-{preset}-s{seed}.bin")
            if synth.PRESETS[preset].n_audio_state >= 1024:
                # medium / large-v3: gigabytes - the oracle's OpenMP generator (proven identical to synth.py tensor by
                # tensor, test_oracle_golden.py) writes the same file in seconds
                from oracle import oracle
                om = oracle.Model.synth(synth.PRESETS[preset].as_list(), seed)
                om.save(path)
                om.close()
            else:
                modelfile.write_synthetic_model(path, synth.PRESETS[preset], seed)
            cache[key] = path
        return cache[key]

    return get
