"""oracle/ — CPU restatement of llmc's weight-quantization hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only
as the checker / the timed CPU baseline. llmc_amd (the product) never imports it and has no CPU fallback.

Pinning: the reference ships no golden vectors or unit tests for this path, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
oracle/make_golden.py (imports /root/reference with two import shims, runs the reference functions on
seeded inputs on CPU) and committed under tests/golden/. tests/test_oracle_golden.py replays them.

Representation: tensors are float32 numpy arrays holding values that are exactly representable in the
logical dtype ('f16' | 'bf16' | 'f32'); every op is evaluated in fp32 and rounded to the op's result
dtype with `rnd`, which is how ATen evaluates the reference's 16-bit elementwise chains.
"""
