"""TEST INFRASTRUCTURE ONLY: CPU oracle for the qgs ensemble tendencies + RK hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  qgs_amd never does (tests/test_layout.py enforces it).
"""
