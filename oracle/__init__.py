"""ORACLE -- test infrastructure only (CPU restatement of the reference's
Newton-step path; see oracle/cones.py header).  Never imported by the product
package under ``conicip.jl_amd/``."""
