# The directory name carries a hyphen (the reference's name); import the host package as
#   sys.path.insert(0, "<repo>/dmft-lanc-ed_amd"); import edgpu
