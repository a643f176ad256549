"""oracle -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's hot-path algorithms.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package heal_amd never imports this.
"""
