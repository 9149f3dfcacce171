"""CPU oracle for the W8A16 hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  ``eetq_amd`` (the product) never does.  See the header of ``eetq_oracle.c`` for the reference
file:line each function restates and for the pinning status.
"""
from .oracle import *  # noqa: F401,F403
