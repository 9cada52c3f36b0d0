# LPVSpectralAMD.jl -- reference-side binding of liblpvspectral.so (include/lpvspectral.h).
#
# Drop-in replacement of the hot path of LPVSpectral.jl (src/lsfft.jl, src/lasso.jl, src/windows.jl): same function
# names, positional arguments and keywords; the arithmetic runs in hand-written HIP kernels on MI355X through @ccall.
# NOT EXECUTABLE in the build image (no julia there).  It is written against the C-ABI, mirrors the Python host code in
# lpvspectral.jl_amd/api.py that the parity tests exercise, and every @ccall below is checked -- symbol, arity and
# argument types -- against the ctypes signature table by tests/test_julia_binding.py.
#
# Conventions used throughout (INTEGRATION.md section 5):
#   * every array handed to the library is first converted to a dense Vector / Matrix of the ABI's eltype, bound to a
#     local name, and that LOCAL is what GC.@preserve protects (never a temporary);
#   * the library never calls back into Julia: progress lines, cb(x,z) and Ctrl-C are handled here between chunks.
module LPVSpectralAMD

using Printf, LinearAlgebra, Statistics
import ProximalOperators                       # the reference's own prox objects are the dispatch types (src/lasso.jl:1)
const PO = ProximalOperators

export ls_spectral, tls_spectral, ls_sparse_spectral, ls_sparse_spectral_lpv, ls_spectral_lpv, ls_windowpsd, ls_windowcsd,
       ls_cohere, ls_windowpsd_lpv, get_fourier_regressor, check_freq, default_freqs, Windows2, Windows3, mapwindows,
       SpectralExt, psd, reshape_params, ADMM, rect, hanning

const LIB = get(ENV, "LPVSPECTRAL_LIB", joinpath(@__DIR__, "..", "lpvspectral.jl_amd", "liblpvspectral.so"))

# ---- status codes -> the exceptions the reference throws --------------------------------------
struct NumericError <: Exception; msg::String; end        # (G + shift I) not positive definite / singular normal equations
last_error() = unsafe_string(@ccall LIB.lpvs_last_error()::Cstring)
function check(rc::Int32)
    rc == 0 && return nothing
    msg = last_error()
    rc == -1 && throw(ArgumentError(msg))                 # src/lsfft.jl:22
    rc == -2 && throw(AssertionError(msg))                # src/lasso.jl:143, src/windows.jl:31
    rc == -3 && throw(DomainError(msg))                   # DSP.arraysplit: noverlap >= n
    rc == -4 && throw(OutOfMemoryError())
    rc == -7 && throw(NumericError(msg))
    error("lpvspectral ($rc): $msg")
end

const EST_SPARSE = Int32(1)                               # LPVS_EST_SPARSE
const EST_DENSE = Int32(2)                                # LPVS_EST_DENSE
const EST_SPARSE_INIT = Int32(3)                          # LPVS_EST_SPARSE_INIT: init = true, x0 = fourier_solve(A, y, zerofreq, λ)

# ---- ProximalOperators objects -> the four device prox kinds -----------------------------------
# Field names are those of ProximalOperators.jl 0.10-0.16 (`lambda`, `r`, `fs`, `idxs`) [PO-recalled, SURVEY.md section 8(c)];
# anything else -- or a sliced sum that is not equal-length contiguous NormL2 groups -- has no device kernel: `nothing`
# makes the callers fall back to the reference's own Julia ADMM.
proxparams(g::PO.NormL1, n) = g.lambda isa Real ? (Int32(1), Float64(g.lambda), Int64(0)) : nothing
proxparams(g::PO.NormL0, n) = (Int32(2), Float64(g.lambda), Int64(0))
proxparams(g::PO.IndBallL0, n) = (Int32(3), Float64(g.r), Int64(0))
function proxparams(g::PO.SlicedSeparableSum, n)          # src/lasso.jl:53-55: Nf groups ((f-1)L+1:fL,), all NormL2(λ)
    length(g.fs) == 1 || return nothing
    fs, idxs = g.fs[1], g.idxs[1]
    (eltype(fs) <: PO.NormL2 && !isempty(fs)) || return nothing
    λ = fs[1].lambda
    all(f -> f.lambda == λ, fs) || return nothing
    L = length(idxs[1][1])
    for (k, ix) in enumerate(idxs)
        (length(ix) == 1 && ix[1] == ((k - 1) * L + 1):(k * L)) || return nothing
    end
    length(idxs) * L <= n || return nothing
    (Int32(4), Float64(λ), Int64(L))
end
proxparams(g, n) = nothing

struct SpectralExt                                         # src/LPVSpectral.jl:59-70
    Y; X; V; w; Nv; λ; coulomb::Bool; normalize::Bool; x; Σ
end
reshape_params(x, Nf) = reshape(x, Nf, :)                  # src/utilities.jl:77
psd(se::SpectralExt) = abs2.(sum(reshape_params(copy(se.x), length(se.w)), dims=2))   # src/lsfft.jl:214-217

default_freqs(n::Int, fs=1) = (0:(n >> 1)) .* (fs / n)     # src/lsfft.jl:3-9 (rfftfreq)
default_freqs(t::AbstractVector, fs=1/mean(diff(t))) = default_freqs(length(t), fs)
default_freqs(t::AbstractVector, n::Int) = default_freqs(t[1:n])

rect(n) = ones(n)                                          # DSP.rect
hanning(n) = n > 1 ? 0.5 .* (1 .+ cos.(2π .* range(-0.5, 0.5, length=n))) : ones(1)   # DSP.hanning

dense(::Type{T}, a) where T = a isa Vector{T} ? a : Vector{T}(vec(collect(a)))

function check_freq(f)                                     # src/lsfft.jl:20-24
    fv = dense(Float64, f); z = Ref{Int64}(0)
    GC.@preserve fv check(@ccall LIB.lpvs_check_freq_f64(fv::Ptr{Float64}, length(fv)::Int64, z::Ref{Int64})::Int32)
    z[] == 0 ? nothing : Int(z[])
end

function get_fourier_regressor(t::AbstractArray{T}, f::AbstractArray{T}) where T   # src/lsfft.jl:26-49
    tv, fv = dense(Float64, t), dense(Float64, f)
    zf = check_freq(fv)
    A = zeros(Float64, length(tv), zf === nothing ? 2length(fv) : 2length(fv) - 1)
    z = Ref{Int64}(0)
    GC.@preserve tv fv A check(@ccall LIB.lpvs_fourier_regressor_f64(tv::Ptr{Float64}, length(tv)::Int64,
        fv::Ptr{Float64}, length(fv)::Int64, A::Ptr{Float64}, z::Ref{Int64})::Int32)
    T.(A), zf
end

# Float32 method (the reference is eltype-generic): same call through the _f32 entry point, Float32 in and out
function get_fourier_regressor(t::AbstractArray{Float32}, f::AbstractArray{Float32})
    tv, fv = dense(Float32, t), dense(Float32, f)
    z = Ref{Int64}(0)
    GC.@preserve fv check(@ccall LIB.lpvs_check_freq_f32(fv::Ptr{Float32}, length(fv)::Int64, z::Ref{Int64})::Int32)
    zf = z[] == 0 ? nothing : Int(z[])
    A = zeros(Float32, length(tv), zf === nothing ? 2length(fv) : 2length(fv) - 1)
    GC.@preserve tv fv A check(@ccall LIB.lpvs_fourier_regressor_f32(tv::Ptr{Float32}, length(tv)::Int64,
        fv::Ptr{Float32}, length(fv)::Int64, A::Ptr{Float32}, z::Ref{Int64})::Int32)
    A, zf
end

# ---- options (include/lpvspectral.h LPVS_OPT_*) ---------------------------------------------------
# Extensions (the reference has none of them): how a handle stores the inverse its ADMM mat-vec streams and how it iterates.
#   storage   = :mixed (>= 36 significant bits per element, all read every iteration; 1e-10 in the iterates) | :split (40 bits) | :f64 (doubles) |
#               :mixed32 (the default of single-signal handles with n >= 2048: :mixed's tiles, of which the iteration reads 32 bits; the 4-bit
#               planes ride, up to 32 iterations stale, in the x-update's offset vector -- x, z, u as with :mixed: include/lpvspectral.h)
#   iteration = :one (default where applicable: one launch per ADMM iteration) | :two
#   gram_form = :ap | :krs | :kr,  nt_loads = :on | :off,  slot_sums = :nufft | :direct        (`nothing` = the library's choice)
# Estimators take them as keywords (`ls_sparse_spectral_lpv(...; storage=:f64)`); results do not depend on them beyond rounding.
#   window_chunk_mb = MB of packed inverses per chunk of the batched-window engine | :uncut;  windows_in_flight = 1 .. 4 parts of a chunk;
#   reserve_cus = CUs the factorisation leaves to its pivot chain | :none                     (integers; defaults only, not handle options)
const OPT_ID = (storage=Int32(1), iteration=Int32(2), gram_form=Int32(3), nt_loads=Int32(4), slot_sums=Int32(5),
                window_chunk_mb=Int32(6), windows_in_flight=Int32(7), reserve_cus=Int32(8), xupdate_correction=Int32(9))
const OPT_VALUES = (storage=(mixed=1, split=2, f64=3, mixed32=4), iteration=(one=1, two=2), gram_form=(ap=1, krs=2, kr=3),
                    nt_loads=(off=1, on=2), slot_sums=(nufft=1, direct=2), window_chunk_mb=(uncut=-1,), windows_in_flight=NamedTuple(),
                    reserve_cus=(none=-1,), xupdate_correction=(on=1, off=2))
optvalue(name::Symbol, v) = v === nothing ? Int32(0) : v isa Integer ? Int32(v) : Int32(getfield(getfield(OPT_VALUES, name), Symbol(v)))
function set_default_option(name::Symbol, v=nothing)       # thread-local: handles created / window batches run afterwards
    oid, vid = getfield(OPT_ID, name), optvalue(name, v)
    check(@ccall LIB.lpvs_set_default_option(oid::Int32, vid::Int32)::Int32)
end
function get_default_option(name::Symbol)
    oid = getfield(OPT_ID, name); r = Ref{Int32}(0)
    check(@ccall LIB.lpvs_get_default_option(oid::Int32, r::Ref{Int32})::Int32)
    r[] == 0 && return nothing
    vals = getfield(OPT_VALUES, name)
    for k in keys(vals); getfield(vals, k) == r[] && return k; end
    Int(r[])                                              # integer-valued options: the number itself
end
const OPTION_KEYS = keys(OPT_ID)
# run f() with the given option keywords as thread defaults, restore the previous defaults afterwards; returns the
# remaining keywords (the reference's own: iters, tol, μ, ...)
function with_options(f, kwargs)
    mine = [(k, v) for (k, v) in pairs(kwargs) if k in OPTION_KEYS && v !== nothing]
    prev = [(k, get_default_option(k)) for (k, _) in mine]
    for (k, v) in mine; set_default_option(k, v); end
    try
        return f((; (k => v for (k, v) in pairs(kwargs) if !(k in OPTION_KEYS))...))
    finally
        for (k, v) in prev; set_default_option(k, v); end
    end
end

# ---- handle wrapper ----------------------------------------------------------------------------
mutable struct Problem
    h::Ptr{Cvoid}; n::Int; m::Int; ns::Int                 # m = complex parameters per signal, ns = right-hand sides
    function Problem(h, m, ns=1)
        n = Ref{Int64}(0); check(@ccall LIB.lpvs_problem_size(h::Ptr{Cvoid}, n::Ref{Int64})::Int32)
        p = new(h, Int(n[]), m, ns)
        finalizer(q -> (@ccall LIB.lpvs_problem_destroy(q.h::Ptr{Cvoid})::Int32), p)
    end
end

function set_option!(p::Problem, name::Symbol, v=nothing)  # one handle; takes effect at the next admm init / run
    oid, vid = getfield(OPT_ID, name), optvalue(name, v)
    check(@ccall LIB.lpvs_problem_set_option(p.h::Ptr{Cvoid}, oid::Int32, vid::Int32)::Int32)
end
function get_option(p::Problem, name::Symbol)              # the value in effect (explicit, thread default or environment)
    oid = getfield(OPT_ID, name); r = Ref{Int32}(0)
    check(@ccall LIB.lpvs_problem_get_option(p.h::Ptr{Cvoid}, oid::Int32, r::Ref{Int32})::Int32)
    r[] == 0 ? nothing : keys(getfield(OPT_VALUES, name))[r[]]
end

function fourier_problem(y, t, f, W; device=0)
    yv, tv, fv = dense(Float64, y), dense(Float64, t), dense(Float64, f)
    @assert length(yv) == length(tv) "y and t has to be the same length"
    Wv = W === nothing ? Float64[] : dense(Float64, W)     # a named local: stays alive under GC.@preserve
    @assert W === nothing || length(Wv) == length(yv) "W has to be the same length as y"
    Wp = W === nothing ? Ptr{Float64}(C_NULL) : pointer(Wv)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve yv tv fv Wv check(@ccall LIB.lpvs_problem_create_fourier_f64(yv::Ptr{Float64}, tv::Ptr{Float64},
        length(yv)::Int64, fv::Ptr{Float64}, length(fv)::Int64, Wp::Ptr{Float64}, Int32(device)::Int32, h::Ref{Ptr{Cvoid}})::Int32)
    Problem(h[], length(fv))
end

function lpv_problem(y, X, V, w, Nv, normalize, coulomb; device=0)
    yv, Xv, Vv, wv = dense(Float64, y), dense(Float64, X), dense(Float64, V), dense(Float64, w)
    @assert length(yv) == length(Xv) == length(Vv) "y, X and V has to be the same length"
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve yv Xv Vv wv check(@ccall LIB.lpvs_problem_create_lpv_f64(yv::Ptr{Float64}, Xv::Ptr{Float64},
        Vv::Ptr{Float64}, length(yv)::Int64, wv::Ptr{Float64}, length(wv)::Int64, Int64(Nv)::Int64, Int32(normalize)::Int32,
        Int32(coulomb)::Int32, Int32(device)::Int32, h::Ref{Ptr{Cvoid}})::Int32)
    Problem(h[], length(wv) * (coulomb ? 2Nv : Nv))
end

# Y is N x ns (one column per signal sharing X, V, w): one Gram, ns right-hand sides
function lpv_multi_problem(Y::AbstractMatrix, X, V, w, Nv, normalize, coulomb; device=0)
    Ym = Matrix{Float64}(Y); Xv, Vv, wv = dense(Float64, X), dense(Float64, V), dense(Float64, w)
    @assert size(Ym, 1) == length(Xv) == length(Vv) "Y, X and V has to have the same number of samples"
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Ym Xv Vv wv check(@ccall LIB.lpvs_problem_create_lpv_multi_f64(Ym::Ptr{Float64}, size(Ym, 2)::Int64,
        Xv::Ptr{Float64}, Vv::Ptr{Float64}, size(Ym, 1)::Int64, wv::Ptr{Float64}, length(wv)::Int64, Int64(Nv)::Int64,
        Int32(normalize)::Int32, Int32(coulomb)::Int32, Int32(device)::Int32, h::Ref{Ptr{Cvoid}})::Int32)
    Problem(h[], length(wv) * (coulomb ? 2Nv : Nv), size(Ym, 2))
end

function dense_problem(A::AbstractMatrix, y; device=0)    # ADMM(x, LeastSquares(A, y; iterative=true), proxg)
    Am, yv = Matrix{Float64}(A), dense(Float64, y)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Am yv check(@ccall LIB.lpvs_problem_create_dense_f64(Am::Ptr{Float64}, yv::Ptr{Float64}, size(Am, 1)::Int64,
        size(Am, 2)::Int64, C_NULL::Ptr{Float64}, Int32(device)::Int32, h::Ref{Ptr{Cvoid}})::Int32)
    Problem(h[], 0)
end
function gram_problem(Q::AbstractMatrix, q; device=0)     # ADMM(x, Quadratic(Q, q; iterative=true), proxg)
    Qm, qv = Matrix{Float64}(Q), dense(Float64, q)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Qm qv check(@ccall LIB.lpvs_problem_create_gram_f64(Qm::Ptr{Float64}, qv::Ptr{Float64}, size(Qm, 1)::Int64,
        Int32(device)::Int32, h::Ref{Ptr{Cvoid}})::Int32)
    Problem(h[], 0)
end

function params(p::Problem, which=0)
    re, im_ = zeros(p.m * p.ns), zeros(p.m * p.ns)
    GC.@preserve re im_ check(@ccall LIB.lpvs_problem_get_params_f64(p.h::Ptr{Cvoid}, Int32(which)::Int32, re::Ptr{Float64}, im_::Ptr{Float64})::Int32)
    complex.(re, im_)
end
function pack(p::Problem, coef::Vector{Float64})
    re, im_ = zeros(p.m), zeros(p.m)
    GC.@preserve coef re im_ check(@ccall LIB.lpvs_problem_pack_params_f64(p.h::Ptr{Cvoid}, coef::Ptr{Float64}, re::Ptr{Float64}, im_::Ptr{Float64})::Int32)
    complex.(re, im_)
end
function solve_ridge(p::Problem, ridge)
    x = zeros(p.n)
    GC.@preserve x check(@ccall LIB.lpvs_problem_solve_ridge_f64(p.h::Ptr{Cvoid}, Float64(ridge)::Float64, x::Ptr{Float64})::Int32)
    x
end
function gram(p::Problem)
    G = zeros(p.n, p.n); b = zeros(p.n)
    GC.@preserve G b check(@ccall LIB.lpvs_problem_get_gram_f64(p.h::Ptr{Cvoid}, G::Ptr{Float64}, b::Ptr{Float64})::Int32)
    G, b
end
function rhs(p::Problem)
    B = zeros(p.n, p.ns)
    GC.@preserve B check(@ccall LIB.lpvs_problem_get_rhs_f64(p.h::Ptr{Cvoid}, B::Ptr{Float64})::Int32)
    B
end
function inverse(p::Problem, shift)
    M = zeros(p.n, p.n)
    GC.@preserve M check(@ccall LIB.lpvs_problem_get_inverse_f64(p.h::Ptr{Cvoid}, Float64(shift)::Float64, M::Ptr{Float64})::Int32)
    M
end
function iterates(p::Problem)
    x, z, u = zeros(p.n, p.ns), zeros(p.n, p.ns), zeros(p.n, p.ns)
    GC.@preserve x z u check(@ccall LIB.lpvs_admm_get_f64(p.h::Ptr{Cvoid}, x::Ptr{Float64}, z::Ptr{Float64}, u::Ptr{Float64})::Int32)
    p.ns == 1 ? (vec(x), vec(z), vec(u)) : (x, z, u)
end
# resume (SURVEY.md section 5): install iterates saved by `iterates` into a freshly initialised handle
function set_state!(p::Problem, x, z, u, iters_done::Integer; offset=nothing)
    xv, zv, uv = dense(Float64, x), dense(Float64, z), dense(Float64, u)
    GC.@preserve xv zv uv check(@ccall LIB.lpvs_admm_set_state_f64(p.h::Ptr{Cvoid}, xv::Ptr{Float64}, zv::Ptr{Float64}, uv::Ptr{Float64},
        Int64(iters_done)::Int64)::Int32)
    if offset !== nothing          # the offset vector saved with the iterates (`offset_vector`): the run continues bit for bit
        ov = dense(Float64, offset)
        GC.@preserve ov check(@ccall LIB.lpvs_admm_set_offset_f64(p.h::Ptr{Cvoid}, ov::Ptr{Float64}, Int64(length(ov))::Int64)::Int32)
    end
end
# the x-update's offset vector currently in effect (handles of n >= 2048; `nothing` otherwise): part of a checkpoint next to `iterates`.
# Opaque: n x ns values, or 2n for a handle that iterates on 32-bit reads of its fixed-point tiles (lpvs_admm_offset_len).
function offset_vector(p::Problem)
    k = Ref{Int64}(0)
    check(@ccall LIB.lpvs_admm_offset_len(p.h::Ptr{Cvoid}, k::Ref{Int64})::Int32)
    k[] == 0 && return nothing
    xb = zeros(k[])
    GC.@preserve xb check(@ccall LIB.lpvs_admm_get_offset_f64(p.h::Ptr{Cvoid}, xb::Ptr{Float64}, Int64(length(xb))::Int64)::Int32)
    (p.ns == 1 || k[] != p.n * p.ns) ? xb : reshape(xb, p.n, p.ns)
end

# ---- ADMM driver: src/lasso.jl:136-171 with the iterations on the GPU ---------------------------
# One blocking @ccall per chunk of `printerval` iterations, so @printf/@info, cb(x,z) and Ctrl-C
# (InterruptException, src/lasso.jl:59-63) are all handled by Julia on the calling thread.
function admm!(p::Problem, x0, prox::Tuple, sign; iters=10000, tol=1e-5, printerval=100, cb=nothing, μ=0.05)
    @assert 0 ≤ μ ≤ 1 "μ should be ≤ 1"                                              # :143
    kind, par, glen = prox
    check(@ccall LIB.lpvs_problem_set_prox(p.h::Ptr{Cvoid}, kind::Int32, par::Float64, glen::Int64)::Int32)
    x0v = x0 === nothing ? Float64[] : dense(Float64, x0)
    x0p = x0 === nothing ? Ptr{Float64}(C_NULL) : pointer(x0v)
    GC.@preserve x0v check(@ccall LIB.lpvs_admm_init_f64(p.h::Ptr{Cvoid}, x0p::Ptr{Float64}, Float64(μ)::Float64,
        Float64(tol)::Float64, Int32(sign)::Int32)::Int32)
    done = 0; conv = false
    it, nxz, cv = Ref{Int64}(0), Ref{Float64}(0), Ref{Int32}(0)
    printerval = printerval > 0 ? printerval : iters
    while done < iters && !conv
        chunk = min(printerval - done % printerval, iters - done)
        check(@ccall LIB.lpvs_admm_run(p.h::Ptr{Cvoid}, Int64(chunk)::Int64, it::Ref{Int64}, nxz::Ref{Float64}, cv::Ref{Int32})::Int32)
        done, conv = Int(it[]), cv[] != 0
        if done % printerval == 0
            @printf("%d ||x-z||₂ %.10f\n", done, nxz[])                              # :159
            cb !== nothing && (xz = iterates(p); cb(xz[1], xz[2]))                   # :160-162
        end
        if conv
            @printf("%d ||x-z||₂ %.10f\n", done, nxz[])                              # :165
            @info("||x-z||₂ ≤ tol")                                                   # :166
        end
    end
    xz = iterates(p)
    xz[1], xz[2]
end

"""`x, z = ADMM(x, proxf, proxg; iters, tol, printerval, cb, μ)` (src/lasso.jl:136-171): `proxf` a
`ProximalOperators.LeastSquares(A, b; iterative=true)` or `Quadratic(Q, q; iterative=true)`, `proxg` one of the four device
prox kinds; anything else is not accelerated (call the reference's ADMM)."""
function ADMM(x::AbstractArray{T}, proxf, proxg; iters=10000, tol=1e-5, printerval=100, cb=nothing, μ=T(0.05)) where T
    kwargs = (iters=iters, tol=tol, printerval=printerval, cb=cb, μ=μ)
    pp = proxparams(proxg, length(x))
    pp === nothing && throw(ArgumentError("proxg of type $(typeof(proxg)) has no device kernel"))
    # field names of the iterative variants [PO-recalled]: LeastSquaresIterative(A, b, lambda...), QuadraticIterative(Q, q)
    if hasproperty(proxf, :A) && hasproperty(proxf, :b)
        x_, z_ = admm!(dense_problem(proxf.A, proxf.b), x, pp, +1; kwargs...)
    elseif hasproperty(proxf, :Q) && hasproperty(proxf, :q)
        x_, z_ = admm!(gram_problem(proxf.Q, proxf.q), x, pp, -1; kwargs...)        # (Q + I/μ) x = -q + v/μ
    else
        throw(ArgumentError("proxf of type $(typeof(proxf)) has no device path"))
    end
    copyto!(x, x_)                                                                    # the reference mutates x in place (:151)
    x, z_
end

# ---- estimators: same signatures as the reference ------------------------------------------------
function ls_spectral(y, t, f=default_freqs(t); λ=1e-10, verbose=false, device=0)     # src/lsfft.jl:62-67
    yv, tv, fv = dense(Float64, y), dense(Float64, t), dense(Float64, f)
    @assert length(yv) == length(tv) "y and t has to be the same length"
    re, im_ = zeros(length(fv)), zeros(length(fv))
    # [A; λI] \ [y; 0] from the device Gram in its better-conditioned form (primal for tall, dual for fat systems such as the
    # default grid of an even-length record, whose Gram is singular): lpvs_ls_spectral_f64, NOT a ridge solve with λ²
    GC.@preserve yv tv fv re im_ check(@ccall LIB.lpvs_ls_spectral_f64(yv::Ptr{Float64}, tv::Ptr{Float64}, length(yv)::Int64,
        fv::Ptr{Float64}, length(fv)::Int64, Float64(λ)::Float64, Int32(device)::Int32, re::Ptr{Float64}, im_::Ptr{Float64})::Int32)
    if verbose                                                                        # :65
        G, _ = gram(fourier_problem(yv, tv, fv, nothing; device=device))
        @info("Condition number: $(round(cond(G), digits=2))\n")
    end
    complex.(re, im_), f
end
function ls_spectral(y, t, f, W::AbstractVector; verbose=false, λ=1e-10, device=0)   # src/lsfft.jl:74-80
    p = fourier_problem(y, t, f, W; device=device)
    x = pack(p, solve_ridge(p, λ))                       # (A'WA + λI) \ A'Wy
    verbose && @info("Condition number: $(round(cond(gram(p)[1]), digits=2))\n")      # :78
    x, f
end

function tls_spectral(y, t, f=default_freqs(t)[1:end-1])                               # src/lsfft.jl:85-99
    p = fourier_problem(y, t, f, nothing)
    G, b = gram(p)
    H = [G b; b' dot(y, y)]                             # [A y]'[A y]: its smallest eigenvector is the last right singular vector
    v = eigen(Symmetric(H)).vectors[:, 1]
    pack(p, -v[1:p.n] ./ v[p.n + 1]), f
end

_x0(q, zf) = zf === nothing ? [real.(q); imag.(q)] : [real.(q); imag.(q[2:end])]     # src/lasso.jl:93-97

function ls_sparse_spectral(y::AbstractArray{T}, t, f=default_freqs(t); init=false, λ=T(1),
                            proxg=PO.NormL1(λ), device=0, kwargs...) where T          # src/lasso.jl:85-102
    with_options(kwargs) do kw                           # storage= / iteration= ... (extensions) apply to the handle created here
        p = fourier_problem(y, t, f, nothing; device=device)
        pp = proxparams(proxg, p.n)
        pp === nothing && throw(ArgumentError("proxg of type $(typeof(proxg)) has no device kernel; use LPVSpectral.ls_sparse_spectral"))
        x0 = init ? _x0(pack(p, solve_ridge(p, λ^2)), check_freq(f)) : nothing       # fourier_solve(A,y,zerofreq,λ), :92
        admm!(p, x0, pp, +1; kw...)
        params(p), f
    end
end
function ls_sparse_spectral(y::AbstractArray{T}, t, f, W; init=false, λ=T(1), proxg=PO.NormL1(T(λ)), device=0,
                            kwargs...) where T                                       # src/lasso.jl:105-126
    with_options(kwargs) do kw
        p = fourier_problem(y, t, f, W; device=device)
        pp = proxparams(proxg, p.n)
        pp === nothing && throw(ArgumentError("proxg of type $(typeof(proxg)) has no device kernel; use LPVSpectral.ls_sparse_spectral"))
        x0 = init ? _x0(ls_spectral(y, t, f; λ=λ, device=device)[1], check_freq(f)) : nothing   # :112 -- the UNWEIGHTED solve
        admm!(p, x0, pp, -1; kw...)                       # Quadratic(Q, q=+A'Wy) as written (:119-121)
        params(p), f
    end
end

function ls_sparse_spectral_lpv(y::AbstractVector{S}, X::AbstractVector{S}, V::AbstractVector{S}, w, Nv::Integer;
                                λ=1, coulomb=false, normalize=true, device=0, kwargs...) where S   # src/lasso.jl:27-70
    coulomb && throw(ArgumentError("coulomb=true is ill-defined in the sparse LPV path (half of x is never written by prox!); use ls_spectral_lpv"))
    w = w[:]
    with_options(kwargs) do kw
        p = lpv_problem(y, X, V, w, Nv, normalize, false; device=device)
        local prm
        try
            admm!(p, nothing, (Int32(4), Float64(λ), Int64(2Nv)), +1; kw...)        # SlicedSeparableSum(NormL2(λ)...), :53-55
            prm = params(p, 0)
        catch e
            e isa InterruptException || rethrow(e)
            @info "Aborting"                              # :61
            prm = params(p, 1)                            # z = copy(x)
        end
        SpectralExt(y, X, V, w, Nv, λ, coulomb, normalize, prm, nothing)
    end
end

# Multichannel batch over several devices (extension; BASELINE.json config 5): the columns of Y (N x ns) share X, V, w; contiguous
# channel ranges go to `ngpus` devices driven by this one process (ngpus = 0: every visible device).  proxg = nothing gives the
# reference's frequency-grouped lasso (src/lasso.jl:53-55) per channel.  Returns a Vector of SpectralExt.
function ls_sparse_spectral_lpv(Y::AbstractMatrix{S}, X::AbstractVector{S}, V::AbstractVector{S}, w, Nv::Integer;
                                λ=1, normalize=true, proxg=nothing, iters=10000, tol=1e-5, μ=0.05, ngpus=0) where S
    @assert 0 ≤ μ ≤ 1 "μ should be ≤ 1"
    w = w[:]
    Ym = Matrix{Float64}(Y); Xv, Vv, wv = dense(Float64, X), dense(Float64, V), dense(Float64, w)
    N, ns = size(Ym); Nf = length(wv); m = Nf * Nv
    @assert N == length(Xv) == length(Vv) "Y, X and V has to have the same number of samples"
    pp = proxg === nothing ? (Int32(4), Float64(λ), Int64(2Nv)) : proxparams(proxg, 2Nf * Nv)
    pp === nothing && throw(ArgumentError("proxg of type $(typeof(proxg)) has no device kernel"))
    re, im_ = zeros(m, ns), zeros(m, ns); its = zeros(Int64, ns)
    GC.@preserve Ym Xv Vv wv re im_ its check(@ccall LIB.lpvs_lpv_batch_multi_f64(Ym::Ptr{Float64}, Int64(ns)::Int64, Xv::Ptr{Float64}, Vv::Ptr{Float64},
        Int64(N)::Int64, wv::Ptr{Float64}, Int64(Nf)::Int64, Int64(Nv)::Int64, Int32(normalize)::Int32, pp[1]::Int32, pp[2]::Float64, pp[3]::Int64,
        Float64(μ)::Float64, Float64(tol)::Float64, Int64(iters)::Int64, C_NULL::Ptr{Int32}, Int32(ngpus)::Int32, re::Ptr{Float64}, im_::Ptr{Float64},
        its::Ptr{Int64})::Int32)
    [SpectralExt(Ym[:, q], X, V, w, Nv, λ, false, normalize, complex.(re[:, q], im_[:, q]), nothing) for q in 1:ns]
end

# Independent signals (extension; BASELINE.json config 3 as a batch): column q of Y, X and V (all N x nsig) is one signal with its own
# samples, i.e. the loop [ls_sparse_spectral_lpv(Y[:,q], X[:,q], V[:,q], w, Nv; ...) for q] inside the library -- contiguous signal
# ranges over `ngpus` devices, `in_flight` solves at a time per device, each on its own handle and stream (the matrix-core-bound Gram /
# factorisation of one solve under the HBM-bound iterations of another).  Returns a Vector of SpectralExt.
function ls_sparse_spectral_lpv(Y::AbstractMatrix{S}, X::AbstractMatrix{S}, V::AbstractMatrix{S}, w, Nv::Integer;
                                λ=1, normalize=true, proxg=nothing, iters=10000, tol=1e-5, μ=0.05, ngpus=0, in_flight=2) where S
    @assert 0 ≤ μ ≤ 1 "μ should be ≤ 1"
    w = w[:]
    Ym, Xm, Vm = Matrix{Float64}(Y), Matrix{Float64}(X), Matrix{Float64}(V); wv = dense(Float64, w)
    N, nsig = size(Ym); Nf = length(wv); m = Nf * Nv
    @assert size(Xm) == size(Vm) == (N, nsig) "Y, X and V has to have the same number of samples"
    pp = proxg === nothing ? (Int32(4), Float64(λ), Int64(2Nv)) : proxparams(proxg, 2Nf * Nv)
    pp === nothing && throw(ArgumentError("proxg of type $(typeof(proxg)) has no device kernel"))
    re, im_ = zeros(m, nsig), zeros(m, nsig); its = zeros(Int64, nsig)
    GC.@preserve Ym Xm Vm wv re im_ its check(@ccall LIB.lpvs_lpv_signals_multi_f64(Ym::Ptr{Float64}, Xm::Ptr{Float64}, Vm::Ptr{Float64},
        Int64(N)::Int64, Int64(nsig)::Int64, wv::Ptr{Float64}, Int64(Nf)::Int64, Int64(Nv)::Int64, Int32(normalize)::Int32, pp[1]::Int32, pp[2]::Float64,
        pp[3]::Int64, Float64(μ)::Float64, Float64(tol)::Float64, Int64(iters)::Int64, C_NULL::Ptr{Int32}, Int32(ngpus)::Int32, Int32(in_flight)::Int32,
        re::Ptr{Float64}, im_::Ptr{Float64}, its::Ptr{Int64})::Int32)
    [SpectralExt(Ym[:, q], Xm[:, q], Vm[:, q], w, Nv, λ, false, normalize, complex.(re[:, q], im_[:, q]), nothing) for q in 1:nsig]
end

function ls_spectral_lpv(Y::AbstractVector, X::AbstractVector, V::AbstractVector, w, Nv::Integer;
                         λ=1e-8, coulomb=false, normalize=true, device=0, covariance=true)   # src/lsfft.jl:239-259
    w = w[:]
    Yv = dense(Float64, Y); N = length(Yv)
    if !covariance                                          # (extension: ls_windowpsd_lpv reads the parameters only -- no Σ, no second inverse)
        p1 = lpv_problem(Yv, X, V, w, Nv, normalize, coulomb; device=device)
        x1 = try solve_ridge(p1, λ^2) catch e; e isa NumericError || rethrow(e); nothing end
        x1 === nothing || return SpectralExt(Y, X, V, w, Nv, λ, coulomb, normalize, pack(p1, x1), nothing)
    end
    p = lpv_multi_problem([Yv ones(N)], X, V, w, Nv, normalize, coulomb; device=device)   # second right-hand side: the constant signal
    Nf, nb = length(w), coulomb ? 2Nv : Nv
    local x
    try
        x = solve_ridge(p, λ^2)                           # real_complex_bs(A, Y, λ) in normal-equation form, permuted column order
    catch e
        e isa NumericError || rethrow(e)                  # (G + λ²I) singular to working precision: the reference's QR route on the host
        Φ = zeros(N, 2Nf * nb)
        Xv, Vv, wv = dense(Float64, X), dense(Float64, V), dense(Float64, w)
        GC.@preserve Xv Vv wv Φ check(@ccall LIB.lpvs_lpv_regressor_f64(Xv::Ptr{Float64}, Vv::Ptr{Float64}, Int64(N)::Int64, wv::Ptr{Float64},
            Int64(Nf)::Int64, Int64(Nv)::Int64, Int32(normalize)::Int32, Int32(coulomb)::Int32, Int32(1)::Int32, Φ::Ptr{Float64})::Int32)
        x = [Φ; λ * I] \ [Yv; zeros(2Nf * nb)]
    end
    prm = pack(p, x)
    G, _ = gram(p); B = rhs(p)
    e2 = dot(x, G * x) - 2dot(B[:, 1], x) + dot(Yv, Yv)   # ‖AA·x − Y‖² from the Gram
    esum = dot(B[:, 2], x) - sum(Yv)                      # Σe through the constant right-hand side
    var_e = (e2 - esum^2 / N) / (N - 1)                   # var(e), :253
    Minv = try inverse(p, λ) catch e; e isa NumericError || rethrow(e); inv(G + λ * I) end   # inv(AA'AA + λI), λ as written
    # reference order u = c·Nf·nb + j·Nf + f  <-  device order q = f·2nb + c·nb + j   (0-based c, j, f)
    perm = Vector{Int}(undef, 2Nf * nb)
    for f in 0:Nf-1, c in 0:1, j in 0:nb-1
        perm[c * Nf * nb + j * Nf + f + 1] = f * 2nb + c * nb + j + 1
    end
    Σ = var_e .* Minv[perm, perm]
    fva = 1 - var_e / var(Yv)                             # :255
    fva < 0.9 && @warn("Fraction of variance explained = $(fva)")                    # :256
    SpectralExt(Y, X, V, w, Nv, λ, coulomb, normalize, prm, Σ)
end

# ---- windows: src/windows.jl (offsets from the C-ABI, views into the arrays) ---------------------
abstract type AbstractWindows end
function _offsets(L::Int, n::Int, noverlap::Int)
    k = Ref{Int64}(0)
    check(@ccall LIB.lpvs_window_count(Int64(L)::Int64, Int64(n)::Int64, Int64(noverlap)::Int64, k::Ref{Int64})::Int32)
    off = zeros(Int64, max(k[], 1))
    GC.@preserve off check(@ccall LIB.lpvs_window_offsets(Int64(L)::Int64, Int64(n)::Int64, Int64(noverlap)::Int64, off::Ptr{Int64},
        Int64(length(off))::Int64, k::Ref{Int64})::Int32)
    off[1:k[]]
end
struct Windows2 <: AbstractWindows; y; t; n::Int; noverlap::Int; W; offsets::Vector{Int64}; end
function Windows2(y::AbstractVector, t::AbstractVector, n::Int=length(y) >> 3, noverlap::Int=n >> 1, window_func=rect)   # :27-36
    noverlap < 0 && (noverlap = n >> 1)
    @assert length(y) == length(t) "y and t has to be the same length"
    Windows2(y, t, n, noverlap, eltype(y).(window_func(n)), _offsets(length(y), n, noverlap))
end
struct Windows3 <: AbstractWindows; y; t; v; n::Int; noverlap::Int; W; offsets::Vector{Int64}; end
function Windows3(y::AbstractVector, t::AbstractVector, v::AbstractVector, n::Int=length(y) >> 3, noverlap::Int=n >> 1, window_func=rect)   # :94-103
    @assert length(y) == length(t) == length(v) "y, t and v has to be the same length"
    noverlap < 0 && (noverlap = n >> 1)
    Windows3(y, t, v, n, noverlap, window_func(n), _offsets(length(y), n, noverlap))
end
Base.length(w::AbstractWindows) = length(w.offsets)
_rng(w, s) = (w.offsets[s] + 1):(w.offsets[s] + w.n)
Base.iterate(w::Windows2, s=1) = s > length(w) ? nothing : ((view(w.y, _rng(w, s)), view(w.t, _rng(w, s))), s + 1)
Base.iterate(w::Windows3, s=1) = s > length(w) ? nothing : ((view(w.y, _rng(w, s)), view(w.t, _rng(w, s)), view(w.v, _rng(w, s))), s + 1)

function merge(yf::AbstractVector{<:AbstractVector}, w::AbstractWindows)              # src/windows.jl:57-70
    flat = zeros(Float64, w.n, length(w))
    for (i, v) in enumerate(yf); flat[:, i] .= v; end    # column i = window i: the ABI's window-major layout
    ym = zeros(length(w.y))
    GC.@preserve flat ym check(@ccall LIB.lpvs_merge_f64(flat::Ptr{Float64}, Int64(length(w))::Int64, Int64(w.n)::Int64, Int64(w.noverlap)::Int64,
        Int64(length(w.y))::Int64, ym::Ptr{Float64})::Int32)
    ym
end
mapwindows(f::Function, W::AbstractWindows) = merge([f(w) for w in W], W)             # :50-53
mapwindows(f::Function, args...) = mapwindows(f, Windows2(args...))

# ---- the batched-window engine: all windows of the drivers below in ONE call ---------------------
# Which estimator / kwargs combinations the engine covers (everything else runs the reference's sequential loop):
#   estimator === ls_spectral         (the 4-argument weighted method, src/lsfft.jl:74-80)    kwargs ⊆ (λ,)
#   estimator === ls_sparse_spectral  (the 4-argument weighted method, src/lasso.jl:105-126)  no cb; init = true and all four device prox kinds included
function engine_args(estimator, nreg; kwargs...)
    kw = Dict{Symbol,Any}(kwargs)
    delete!(kw, :device)
    for k in OPTION_KEYS; delete!(kw, k); end                # (applied as thread defaults by the drivers)
    if estimator === ls_spectral
        (get(kw, :verbose, false) || !issubset(keys(kw), (:λ, :verbose))) && return nothing
        return (est=EST_DENSE, lam=Float64(get(kw, :λ, 1e-10)), prox=(Int32(1), 0.0, Int64(0)), μ=0.05, tol=0.0, iters=0, sign=Int32(1))
    elseif estimator === ls_sparse_spectral
        get(kw, :cb, nothing) !== nothing && return nothing                          # a per-iteration callback needs the host loop
        issubset(keys(kw), (:λ, :proxg, :μ, :tol, :iters, :printerval, :cb, :init)) || return nothing
        g = get(kw, :proxg, PO.NormL1(Float64(get(kw, :λ, 1.0))))
        pp = proxparams(g, nreg)
        pp === nothing && return nothing
        μ = Float64(get(kw, :μ, 0.05))
        @assert 0 ≤ μ ≤ 1 "μ should be ≤ 1"
        init = Bool(get(kw, :init, false))                                           # src/lasso.jl:112: one batched ridge solve in the engine
        return (est=init ? EST_SPARSE_INIT : EST_SPARSE, lam=init ? Float64(get(kw, :λ, 1.0)) : 0.0, prox=pp, μ=μ,
                tol=Float64(get(kw, :tol, 1e-5)), iters=Int(get(kw, :iters, 10000)), sign=Int32(-1))
    end
    nothing
end

# x[Nf, k, ns] complex: column (·, i, s) = fourier2complex of window i of signal s; `ngpus` devices are driven by this one
# process (contiguous window ranges per device, one RCCL all-gather of the coefficients; ngpus = 0: every visible device)
function windows_estimate(Ys::Vector, t, freqs, n::Int, noverlap::Int, W, eng; ngpus::Int=1)
    L = length(Ys[1]); ns = length(Ys)
    Ym = Matrix{Float64}(undef, L, ns)
    for (s, y) in enumerate(Ys); Ym[:, s] .= y; end
    tv, fv, Wv = dense(Float64, t), dense(Float64, freqs), dense(Float64, W)
    k = length(_offsets(L, n, noverlap)); Nf = length(fv)
    re, im_ = zeros(Nf, k, ns), zeros(Nf, k, ns)          # column-major (Nf, k, ns) == the ABI's ns x k x Nf
    its = zeros(Int64, k, ns)
    GC.@preserve Ym tv fv Wv re im_ its check(@ccall LIB.lpvs_windows_estimate_multi_f64(Ym::Ptr{Float64}, Int64(ns)::Int64, tv::Ptr{Float64},
        Int64(L)::Int64, Int64(n)::Int64, Int64(noverlap)::Int64, Wv::Ptr{Float64}, fv::Ptr{Float64}, Int64(Nf)::Int64, eng.est::Int32,
        eng.lam::Float64, eng.prox[1]::Int32, eng.prox[2]::Float64, eng.prox[3]::Int64, eng.μ::Float64, eng.tol::Float64,
        Int64(eng.iters)::Int64, eng.sign::Int32, C_NULL::Ptr{Int32}, Int32(ngpus)::Int32, re::Ptr{Float64}, im_::Ptr{Float64},
        its::Ptr{Int64})::Int32)
    complex.(re, im_), its
end

# The raw ADMM state of every window instead of the packed coefficients: (x, z) as `ADMM` returns them (src/lasso.jl:170) and the scaled dual
# variable u, each Nreg x k x ns (Nreg = 2Nf, or 2Nf - 1 with the zero frequency first; the reference's ordering [re; im], src/lasso.jl:93-97),
# and the iteration counts.  Sparse estimators, one device.  (What the parity tests hold the window batches' iteration to the CPU reference with.)
function windows_estimate_state(Ys::Vector, t, freqs, n::Int, noverlap::Int, W, eng; device::Int=0)
    L = length(Ys[1]); ns = length(Ys)
    Ym = Matrix{Float64}(undef, L, ns)
    for (s, y) in enumerate(Ys); Ym[:, s] .= y; end
    tv, fv, Wv = dense(Float64, t), dense(Float64, freqs), dense(Float64, W)
    k = length(_offsets(L, n, noverlap)); Nf = length(fv)
    nreg = check_freq(fv) === nothing ? 2Nf : 2Nf - 1
    x, z, u = zeros(nreg, k, ns), zeros(nreg, k, ns), zeros(nreg, k, ns)     # column-major (Nreg, k, ns) == the ABI's ns x k x Nreg
    its = zeros(Int64, k, ns)
    GC.@preserve Ym tv fv Wv x z u its check(@ccall LIB.lpvs_windows_estimate_state_f64(Ym::Ptr{Float64}, Int64(ns)::Int64, tv::Ptr{Float64},
        Int64(L)::Int64, Int64(n)::Int64, Int64(noverlap)::Int64, Wv::Ptr{Float64}, fv::Ptr{Float64}, Int64(Nf)::Int64, eng.est::Int32,
        eng.lam::Float64, eng.prox[1]::Int32, eng.prox[2]::Float64, eng.prox[3]::Int64, eng.μ::Float64, eng.tol::Float64,
        Int64(eng.iters)::Int64, eng.sign::Int32, Int64(0)::Int64, Int64(k)::Int64, Int32(device)::Int32, x::Ptr{Float64}, z::Ptr{Float64},
        u::Ptr{Float64}, its::Ptr{Int64})::Int32)
    x, z, u, its
end

function ls_windowpsd(y, t, freqs=nothing; nw=8, noverlap=-1, window_func=rect, estimator=ls_spectral, ngpus=1, kwargs...)
    n = length(y) ÷ nw                                                               # src/lsfft.jl:112-126
    freqs === nothing && (freqs = default_freqs(t, n))
    windows = Windows2(y, t, n, noverlap, window_func)
    nw = length(windows)                                                             # :116 (recomputed)
    S = zeros(eltype(y), length(freqs))
    eng = nw > 0 ? engine_args(estimator, 2length(freqs); kwargs...) : nothing
    if eng !== nothing
        x, _ = with_options(kwargs) do _; windows_estimate([y], t, freqs, n, windows.noverlap, windows.W, eng; ngpus=ngpus); end
        for i in 1:nw
            S .+= abs2.(view(x, :, i, 1))                                            # :122, window order
        end
        return S ./ nw^2, freqs                                                      # :125
    end
    for (yi, ti) in windows
        x = estimator(yi, ti, freqs, windows.W; kwargs...)[1]                        # :121
        S .+= abs2.(x)
    end
    S ./ nw^2, freqs
end

function ls_windowcsd(y, u, t, freqs=nothing; nw=10, noverlap=-1, window_func=rect, estimator=ls_spectral, ngpus=1, kwargs...)
    n = length(y) ÷ nw                                                               # src/lsfft.jl:140-156
    freqs === nothing && (freqs = default_freqs(t, n))
    S = zeros(ComplexF64, length(freqs))
    windowsy = Windows2(y, t, n, noverlap, window_func)
    windowsu = Windows2(u, t, n, noverlap, window_func)
    nw = length(windowsy)
    eng = nw > 0 ? engine_args(estimator, 2length(freqs); kwargs...) : nothing
    if eng !== nothing                                    # one Gram and one factorisation per window serve both signals
        x, _ = with_options(kwargs) do _; windows_estimate([y, u], t, freqs, n, windowsy.noverlap, windowsy.W, eng; ngpus=ngpus); end
        for i in 1:nw
            S += view(x, :, i, 1) .* conj.(view(x, :, i, 2))                         # :152
        end
        return S ./ nw, freqs
    end
    for ((yi, ti), (ui, _)) in zip(windowsy, windowsu)
        xy = estimator(yi, ti, freqs, windowsy.W; kwargs...)[1]
        xu = estimator(ui, ti, freqs, windowsu.W; kwargs...)[1]
        S += xy .* conj.(xu)
    end
    S ./ nw, freqs
end

function ls_cohere(y, u, t, freqs=nothing; nw=10, noverlap=-1, estimator=ls_spectral, ngpus=1, kwargs...)
    n = length(y) ÷ nw                                                               # src/lsfft.jl:176-193
    freqs === nothing && (freqs = default_freqs(t, n))
    Syy, Suu = zeros(length(freqs)), zeros(length(freqs))
    Syu = zeros(ComplexF64, length(freqs))
    windows = Windows3(y, t, u, n, noverlap, hanning)                                # :182
    eng = length(windows) > 0 ? engine_args(estimator, 2length(freqs); kwargs...) : nothing
    if eng !== nothing
        x, _ = with_options(kwargs) do _; windows_estimate([y, u], t, freqs, n, windows.noverlap, windows.W, eng; ngpus=ngpus); end
        for i in 1:length(windows)
            xy, xu = view(x, :, i, 1), view(x, :, i, 2)
            Syu .+= xy .* conj.(xu); Syy .+= abs2.(xy); Suu .+= abs2.(xu)           # :187-189
        end
        return abs2.(Syu) ./ (Suu .* Syy), freqs
    end
    for (yi, ti, ui) in windows
        xy = estimator(yi, ti, freqs, windows.W; kwargs...)[1]
        xu = estimator(ui, ti, freqs, windows.W; kwargs...)[1]
        Syu .+= xy .* conj.(xu); Syy .+= abs2.(xy); Suu .+= abs2.(xu)
    end
    abs2.(Syu) ./ (Suu .* Syy), freqs
end

# The windows' regressors share nothing, so every window is a device solve of its own; `in_flight` of them (an extension; needs
# `julia -t N`) run concurrently, each on its own handle and stream -- statically scheduled, so a solve stays on the thread whose
# thread-local option defaults and error string it uses (defaults set with `set_default_option` on the calling thread do not reach the
# others: pass options as keywords).  The sum is taken in window order either way (src/lsfft.jl:274).
function ls_windowpsd_lpv(Y::AbstractVector, X::AbstractVector, V::AbstractVector, w, Nv::Integer, nw::Int=10, noverlap=0; in_flight::Int=4, kwargs...)
    S = zeros(length(w))                                                             # src/lsfft.jl:267-277
    # the library's own driver (the same device solves on the library's worker threads: no `julia -t` needed) whenever the call has only
    # what it takes; a singular window (NumericError) sends the call down the per-window path below, which has the host-QR route
    if all(k in (:λ, :coulomb, :normalize, :device) for k in keys(kwargs))
        @assert length(Y) == length(X) == length(V) "y, t and v has to be the same length"
        Yv, Xv, Vv, wv = dense(Float64, Y), dense(Float64, X), dense(Float64, V), dense(Float64, w[:])
        kw = (; kwargs...)
        kcnt = Ref{Int64}(0)
        check(@ccall LIB.lpvs_window_count(length(Yv)::Int64, Int64(length(Yv) ÷ nw)::Int64, Int64(noverlap)::Int64, kcnt::Ref{Int64})::Int32)
        fva = ones(max(kcnt[], 1))
        try
            GC.@preserve Yv Xv Vv wv S fva check(@ccall LIB.lpvs_windowpsd_lpv_f64(Yv::Ptr{Float64}, Xv::Ptr{Float64}, Vv::Ptr{Float64}, length(Yv)::Int64,
                wv::Ptr{Float64}, length(wv)::Int64, Int64(Nv)::Int64, Int64(length(Yv) ÷ nw)::Int64, Int64(noverlap)::Int64, Float64(get(kw, :λ, 1e-8))::Float64,
                Int32(get(kw, :normalize, true))::Int32, Int32(get(kw, :coulomb, false))::Int32, Int32(get(kw, :device, 0))::Int32,
                Int32(clamp(in_flight, 1, 8))::Int32, S::Ptr{Float64}, fva::Ptr{Float64})::Int32)
            for v in fva[1:kcnt[]]
                v < 0.9 && @warn("Fraction of variance explained = $(v)")            # src/lsfft.jl:255-256, per window
            end
            return S
        catch e
            e isa NumericError || rethrow()
            fill!(S, 0.0)
        end
    end
    windows = collect(Windows3(Y, X, V, length(Y) ÷ nw, noverlap, rect))
    solve((y, x, v)) = ls_spectral_lpv(collect(y), collect(x), collect(v), w, Nv; covariance=false, kwargs...)
    ses = Vector{Any}(undef, length(windows))
    if in_flight > 1 && Threads.nthreads() > 1 && length(windows) > 1
        nt = min(in_flight, Threads.nthreads())
        Threads.@threads :static for k in 1:nt
            for i in k:nt:length(windows)
                ses[i] = solve(windows[i])
            end
        end
    else
        for i in eachindex(windows)
            ses[i] = solve(windows[i])
        end
    end
    for se in ses
        S += vec(abs2.(sum(reshape_params(se.x, length(w)), dims=2)))
    end
    S
end

end # module
