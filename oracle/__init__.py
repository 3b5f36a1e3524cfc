"""CPU oracle for the VM-ASR hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (vm_asr_amd) never does; it fails loudly without its HIP library.
"""
from .oracle import *  # noqa: F401,F403
