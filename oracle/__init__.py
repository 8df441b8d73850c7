"""TEST INFRASTRUCTURE ONLY.  CPU restatements of the reference's algorithm, used as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under diffusionremotesensing_amd/
imports this package."""
