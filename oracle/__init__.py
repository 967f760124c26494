"""CPU restatement of the reference's hot path - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product path (multiposenet_amd) never does and has no CPU fallback.
"""
