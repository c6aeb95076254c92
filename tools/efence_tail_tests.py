"""(helper for tools/efence_run.py) the tests of tests/test_hip_parity.py BEHIND test_block_L8192_fp32_vs_torch_reference -- where a leaking
fence allocator ran out of resources in one pass over the file -- minus the full-size / determinism / graph ones"""
import subprocess, sys, pytest
ids = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hip_parity.py", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"],
                     capture_output=True, text=True).stdout.splitlines()
ids = [i for i in ids if "::" in i]
cut = max(k for k, i in enumerate(ids) if "test_block_L8192" in i)
sel = [i for i in ids[cut + 1:] if not any(w in i for w in ("full_size", "deterministic", "graph", "L8192", "10000"))]
print(len(ids), "collected,", len(sel), "selected behind the L8192 test", flush=True)
sys.exit(pytest.main(sel + ["-q", "-p", "no:cacheprovider", "-x"]))
