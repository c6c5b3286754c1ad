"""mdx: MI355X-native MD engine standing in for the `gmx mdrun` stages of the
ERa-estradiol Snakemake pipeline (/root/reference/Snakefile:705-707, 738-740, 991-998).

The directory name carries a hyphen, so import it through the root-level alias module
`mdx` (``import mdx``) or ``importlib.import_module("molecular-dynamics_amd")``.
"""
__version__ = "0.1.0"
