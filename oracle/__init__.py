"""oracle/ -- CPU restatement of the reference's algorithm for the APE-L_D forward pass.

TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
nothing under ape_amd/ does.  See oracle/README.md for how the restatement is pinned to the reference.
"""
