"""Residue-name tables used when parsing / writing CA-only PDB records.

Same mapping the reference uses (programs/Foldclass/constants.py:1-10): the 20 standard
residues, UNK -> X, and the protonation-state aliases of ASP/GLU/HIS/LYS.
"""
_STANDARD = "ALA:A CYS:C ASP:D GLU:E PHE:F GLY:G HIS:H ILE:I LYS:K LEU:L MET:M ASN:N PRO:P GLN:Q ARG:R SER:S THR:T VAL:V TRP:W TYR:Y UNK:X"
_ALIASES = "ASH:D GLH:E HID:H HIE:H HIP:H HSD:H HSE:H LYN:K"

three_to_single_aa = {pair.split(":")[0]: pair.split(":")[1] for pair in (_STANDARD + " " + _ALIASES).split()}
# one-letter -> three-letter: the LAST three-letter name listed for a letter wins, as in the
# reference's dict inversion (so D -> ASH, E -> GLH, H -> HSE, K -> LYN)
single_to_three_aa = {one: three for three, one in three_to_single_aa.items()}
