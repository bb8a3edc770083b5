"""TEST INFRASTRUCTURE ONLY.

`oracle/` is a CPU fp32 restatement of the reference's algorithm for the Versatile-Diffusion sampling
path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
package (versatile-diffusion_amd/) never does.
"""
