"""TEST INFRASTRUCTURE — the CPU oracle (checker) for ballista_amd.

PARITY UNPINNED BY THE REFERENCE: kyprifog/ballista's executor hot path is DataFusion /
arrow-rs 4.0.0-SNAPSHOT (git rev 46161d2, rust/Cargo.lock:77-80,497-500), which is not
vendored, cannot be built here (no Rust toolchain) and is exercised by no result-checking
test in the reference.  This package restates the operator semantics
of SURVEY.md Appendix A on the CPU (numpy + a small C library) and is pinned by the
reference's 20-row lineitem fixture plus fsum / pyarrow-Acero cross-checks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (ballista_amd/) never does, and fails loudly without its HIP library.
"""
