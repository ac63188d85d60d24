# tools/make_reference_fixture.jl -- for a MAINTAINER with Julia 0.6 and a checkout of una-dinosauria/local-search-quantization.
#
# Pins this repository's oracle against the REFERENCE ITSELF (VERDICT r1: "parity unpinned" -- no Julia in the build image, the
# reference has no tests or golden vectors).  It calls the reference's own functions on seeded inputs and dumps inputs + outputs to
# one little-endian binary file that tests/test_reference_fixture.py loads when it finds it under tests/golden/ref_*.bin:
#
#     cd <reference checkout>
#     julia <this repo>/tools/make_reference_fixture.jl <this repo>/tests/golden/ref_d32_m8.bin 32 400 8 1234
#
# What is dumped (all from the reference's code, nothing from this repository):
#     X (d x n Float32), C (m codebooks d x 256), B0 (m x n Int16, 1-based)
#     unaries  = get_unaries(X, C)                               src/utils.jl:94-122     m x (256 x n)
#     binaries = get_binaries(C)                                 src/utils.jl:125-144    ncbi x (256 x 256), cbi 2 x ncbi
#     cost0    = veccost(X, B0, C)                               src/utils.jl:225-254    n
#     B1 after encode_icm_fully!(B, X, C, binaries, cbi, 4, false, 0, 1:n, false)   src/encodings/encode_icm.jl:4-127
#              niter = 4 sweeps, randord = false, npert = 0: NO random numbers are drawn, so the call is deterministic and its
#              codes are comparable with oracle/lsq_oracle.c (up to BLAS summation order in the tables: the test uses margins)
#     cost1    = veccost(X, B1, C)
# File layout: magic "LSQREF01", Int32 d, n, m, h, ncbi; then the arrays in the order above, column-major, raw.
include("src/utils.jl")
include("src/encodings/encode_icm.jl")

function main(args)
  out  = args[1]
  d    = parse(Int, args[2]); n = parse(Int, args[3]); m = parse(Int, args[4]); seed = parse(Int, args[5])
  h    = 256
  srand(seed)
  # SIFT-like integer-valued data; codebooks = sampled data vectors / m (the synthetic set-up of this repository's tests)
  X  = convert(Matrix{Float32}, floor.(rand(Float32, d, n) * 256f0))
  C  = Vector{Matrix{Float32}}(m)
  for j = 1:m
    C[j] = convert(Matrix{Float32}, floor.(rand(Float32, d, h) * 256f0) / Float32(m))
  end
  B0 = convert(Matrix{Int16}, rand(1:h, m, n))

  unaries       = get_unaries(X, C, false)
  binaries, cbi = get_binaries(C)
  cost0         = veccost(X, B0, C)
  B1            = copy(B0)
  encode_icm_fully!(B1, X, C, binaries, cbi, 4, false, 0, 1:n, false)
  cost1         = veccost(X, B1, C)

  open(out, "w") do f
    write(f, "LSQREF01")
    write(f, Int32(d)); write(f, Int32(n)); write(f, Int32(m)); write(f, Int32(h)); write(f, Int32(length(binaries)))
    write(f, X)
    for j = 1:m; write(f, C[j]); end
    write(f, B0)
    for j = 1:m; write(f, unaries[j]); end
    for i = 1:length(binaries); write(f, binaries[i]); end
    write(f, convert(Matrix{Int32}, cbi))
    write(f, convert(Vector{Float32}, cost0))
    write(f, B1)
    write(f, convert(Vector{Float32}, cost1))
  end
  println("wrote $out: d=$d n=$n m=$m ncbi=$(length(binaries))")
end

main(ARGS)
