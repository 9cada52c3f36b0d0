# LPVSpectralAMD.jl -- reference-side binding of liblpvspectral.so (include/lpvspectral.h).
#
# Drop-in replacement of the hot path of LPVSpectral.jl (src/lsfft.jl, src/lasso.jl): same function
# names, positional arguments and keywords; the arithmetic runs in hand-written HIP kernels on an
# MI355X through @ccall.  NOT EXECUTABLE in the build image (no julia there); it is written against
# the C-ABI and mirrors, line for line, the Python host code in lpvspectral.jl_amd/api.py that the
# parity tests exercise.
module LPVSpectralAMD

using Printf, LinearAlgebra

export ls_spectral, tls_spectral, ls_sparse_spectral, ls_sparse_spectral_lpv, ls_spectral_lpv, ls_windowpsd,
       get_fourier_regressor, check_freq, default_freqs, Windows2, SpectralExt, psd,
       NormL1, NormL0, IndBallL0, GroupL2

const LIB = get(ENV, "LPVSPECTRAL_LIB", joinpath(@__DIR__, "..", "lpvspectral.jl_amd", "liblpvspectral.so"))

# ---- status codes -> the exceptions the reference throws --------------------------------------
last_error() = unsafe_string(@ccall LIB.lpvs_last_error()::Cstring)
function check(rc::Int32)
    rc == 0 && return
    msg = last_error()
    rc == -1 && throw(ArgumentError(msg))                 # src/lsfft.jl:22
    rc == -2 && throw(AssertionError(msg))                # src/lasso.jl:143
    rc == -3 && throw(DomainError(msg))                   # DSP.arraysplit
    rc == -4 && throw(OutOfMemoryError())
    error("lpvspectral ($rc): $msg")
end

# ---- prox objects (parameters only; ProximalOperators types can be mapped onto these) ---------
struct NormL1;    λ::Float64; end
struct NormL0;    λ::Float64; end
struct IndBallL0; r::Int;     end
struct GroupL2;   λ::Float64; len::Int; end               # SlicedSeparableSum(NormL2(λ)...), src/lasso.jl:53-55
proxparams(g::NormL1) = (Int32(1), g.λ, 0)
proxparams(g::NormL0) = (Int32(2), g.λ, 0)
proxparams(g::IndBallL0) = (Int32(3), Float64(g.r), 0)
proxparams(g::GroupL2) = (Int32(4), g.λ, g.len)

struct SpectralExt                                         # src/LPVSpectral.jl:59-70
    Y; X; V; w; Nv; λ; coulomb::Bool; normalize::Bool; x; Σ
end
psd(se::SpectralExt) = abs2.(sum(reshape(copy(se.x), length(se.w), :), dims=2))   # src/lsfft.jl:214-217

default_freqs(n::Int, fs=1) = (0:(n >> 1)) .* (fs / n)     # src/lsfft.jl:3-9 (rfftfreq)
default_freqs(t::AbstractVector, fs=1 / (sum(diff(t)) / (length(t) - 1))) = default_freqs(length(t), fs)
default_freqs(t::AbstractVector, n::Int) = default_freqs(t[1:n])

function check_freq(f)                                     # src/lsfft.jl:20-24
    fv = Vector{Float64}(f); z = Ref{Int64}(0)
    check(@ccall LIB.lpvs_check_freq_f64(fv::Ptr{Float64}, length(fv)::Int64, z::Ref{Int64})::Int32)
    z[] == 0 ? nothing : Int(z[])
end

function get_fourier_regressor(t::AbstractArray{T}, f::AbstractArray{T}) where T   # src/lsfft.jl:26-49
    tv, fv = Vector{Float64}(t), Vector{Float64}(f)
    zf = check_freq(fv)
    A = zeros(Float64, length(tv), zf === nothing ? 2length(fv) : 2length(fv) - 1)
    z = Ref{Int64}(0)
    GC.@preserve tv fv A check(@ccall LIB.lpvs_fourier_regressor_f64(tv::Ptr{Float64}, length(tv)::Int64,
        fv::Ptr{Float64}, length(fv)::Int64, A::Ptr{Float64}, z::Ref{Int64})::Int32)
    A, zf
end

# Float32 method (the reference is eltype-generic): same call through the _f32 entry point, Float32 in and out
function get_fourier_regressor(t::AbstractArray{Float32}, f::AbstractArray{Float32})
    tv, fv = Vector{Float32}(t), Vector{Float32}(f)
    z = Ref{Int64}(0)
    check(@ccall LIB.lpvs_check_freq_f32(fv::Ptr{Float32}, length(fv)::Int64, z::Ref{Int64})::Int32)
    zf = z[] == 0 ? nothing : Int(z[])
    A = zeros(Float32, length(tv), zf === nothing ? 2length(fv) : 2length(fv) - 1)
    GC.@preserve tv fv A check(@ccall LIB.lpvs_fourier_regressor_f32(tv::Ptr{Float32}, length(tv)::Int64,
        fv::Ptr{Float32}, length(fv)::Int64, A::Ptr{Float32}, z::Ref{Int64})::Int32)
    A, zf
end
# (lpvs_problem_create_{fourier,lpv}_f32, lpvs_admm_init_f32, lpvs_admm_get_f32, lpvs_problem_get_params_f32 and
#  lpvs_ls_spectral_f32 bind the same way; handles created through them stream a single-precision copy of the matrix
#  in the ADMM mat-vec.)

# ---- handle wrapper ----------------------------------------------------------------------------
mutable struct Problem
    h::Ptr{Cvoid}; n::Int; m::Int                          # m = number of complex parameters
    function Problem(h, m)
        n = Ref{Int64}(0); check(@ccall LIB.lpvs_problem_size(h::Ptr{Cvoid}, n::Ref{Int64})::Int32)
        p = new(h, Int(n[]), m)
        finalizer(q -> (@ccall LIB.lpvs_problem_destroy(q.h::Ptr{Cvoid})::Int32), p)
    end
end

function fourier_problem(y, t, f, W; device=0)
    yv, tv, fv = Vector{Float64}(y), Vector{Float64}(t), Vector{Float64}(f)
    @assert length(yv) == length(tv) "y and t has to be the same length"
    Wv = W === nothing ? C_NULL : pointer(Vector{Float64}(W))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve yv tv fv W check(@ccall LIB.lpvs_problem_create_fourier_f64(yv::Ptr{Float64}, tv::Ptr{Float64},
        length(yv)::Int64, fv::Ptr{Float64}, length(fv)::Int64, Wv::Ptr{Float64}, device::Int32, h::Ref{Ptr{Cvoid}})::Int32)
    Problem(h[], length(fv))
end

function lpv_problem(y, X, V, w, Nv, normalize, coulomb; device=0)
    yv, Xv, Vv, wv = Vector{Float64}(y), Vector{Float64}(X), Vector{Float64}(V), Vector{Float64}(w[:])
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve yv Xv Vv wv check(@ccall LIB.lpvs_problem_create_lpv_f64(yv::Ptr{Float64}, Xv::Ptr{Float64},
        Vv::Ptr{Float64}, length(yv)::Int64, wv::Ptr{Float64}, length(wv)::Int64, Nv::Int64, normalize::Int32,
        coulomb::Int32, device::Int32, h::Ref{Ptr{Cvoid}})::Int32)
    Problem(h[], length(wv) * (coulomb ? 2Nv : Nv))
end

function params(p::Problem, which=0)
    re, im_ = zeros(p.m), zeros(p.m)
    check(@ccall LIB.lpvs_problem_get_params_f64(p.h::Ptr{Cvoid}, which::Int32, re::Ptr{Float64}, im_::Ptr{Float64})::Int32)
    complex.(re, im_)
end
function pack(p::Problem, coef)
    re, im_ = zeros(p.m), zeros(p.m)
    check(@ccall LIB.lpvs_problem_pack_params_f64(p.h::Ptr{Cvoid}, coef::Ptr{Float64}, re::Ptr{Float64}, im_::Ptr{Float64})::Int32)
    complex.(re, im_)
end
function solve_ridge(p::Problem, ridge)
    x = zeros(p.n)
    check(@ccall LIB.lpvs_problem_solve_ridge_f64(p.h::Ptr{Cvoid}, Float64(ridge)::Float64, x::Ptr{Float64})::Int32)
    x
end
function iterates(p::Problem)
    x, z = zeros(p.n), zeros(p.n)
    check(@ccall LIB.lpvs_admm_get_f64(p.h::Ptr{Cvoid}, x::Ptr{Float64}, z::Ptr{Float64}, C_NULL::Ptr{Float64})::Int32)
    x, z
end

# ---- ADMM driver: src/lasso.jl:136-171 with the iterations on the GPU ---------------------------
# One blocking @ccall per chunk of `printerval` iterations, so @printf/@info, cb(x,z) and Ctrl-C
# (InterruptException, src/lasso.jl:59-63) are all handled by Julia on the calling thread.
function admm!(p::Problem, x0, proxg, sign; iters=10000, tol=1e-5, printerval=100, cb=nothing, μ=0.05)
    @assert 0 ≤ μ ≤ 1 "μ should be ≤ 1"
    kind, par, glen = proxparams(proxg)
    check(@ccall LIB.lpvs_problem_set_prox(p.h::Ptr{Cvoid}, kind::Int32, par::Float64, glen::Int64)::Int32)
    x0p = x0 === nothing ? C_NULL : pointer(x0)
    GC.@preserve x0 check(@ccall LIB.lpvs_admm_init_f64(p.h::Ptr{Cvoid}, x0p::Ptr{Float64}, Float64(μ)::Float64,
        Float64(tol)::Float64, Int32(sign)::Int32)::Int32)
    done = 0; conv = false
    it, nxz, cv = Ref{Int64}(0), Ref{Float64}(0), Ref{Int32}(0)
    while done < iters && !conv
        chunk = min(printerval - done % printerval, iters - done)
        check(@ccall LIB.lpvs_admm_run(p.h::Ptr{Cvoid}, chunk::Int64, it::Ref{Int64}, nxz::Ref{Float64}, cv::Ref{Int32})::Int32)
        done, conv = Int(it[]), cv[] != 0
        if done % printerval == 0
            @printf("%d ||x-z||₂ %.10f\n", done, nxz[])
            cb !== nothing && cb(iterates(p)...)
        end
        if conv
            @printf("%d ||x-z||₂ %.10f\n", done, nxz[])
            @info("||x-z||₂ ≤ tol")
        end
    end
    iterates(p)
end

# ---- estimators: same signatures as the reference ------------------------------------------------
function ls_spectral(y, t, f=default_freqs(t); λ=1e-10, verbose=false)              # src/lsfft.jl:62-67
    p = fourier_problem(y, t, f, nothing)
    pack(p, solve_ridge(p, λ^2)), f                       # [A; λI] \ [y; 0] in normal-equation form
end
function ls_spectral(y, t, f, W::AbstractVector; verbose=false, λ=1e-10)             # src/lsfft.jl:74-80
    p = fourier_problem(y, t, f, W)
    pack(p, solve_ridge(p, λ)), f                         # (A'WA + λI) \ A'Wy
end

function tls_spectral(y, t, f=default_freqs(t)[1:end-1])                               # src/lsfft.jl:85-99
    p = fourier_problem(y, t, f, nothing)
    G = zeros(p.n, p.n); b = zeros(p.n)
    check(@ccall LIB.lpvs_problem_get_gram_f64(p.h::Ptr{Cvoid}, G::Ptr{Float64}, b::Ptr{Float64})::Int32)
    H = [G b; b' dot(y, y)]                             # [A y]'[A y]: its smallest eigenvector is the last right singular vector
    v = eigen(Symmetric(H)).vectors[:, 1]
    pack(p, -v[1:p.n] ./ v[p.n + 1]), f
end

function ls_sparse_spectral(y::AbstractArray{T}, t, f=default_freqs(t); init=false, λ=T(1),
                            proxg=NormL1(λ), kwargs...) where T                      # src/lasso.jl:85-102
    p = fourier_problem(y, t, f, nothing)
    x0 = nothing
    if init
        q = pack(p, solve_ridge(p, λ^2)); zf = check_freq(f)
        x0 = zf === nothing ? [real.(q); imag.(q)] : [real.(q); imag.(q[2:end])]
    end
    admm!(p, x0, proxg, +1; kwargs...)
    params(p), f
end
function ls_sparse_spectral(y::AbstractArray{T}, t, f, W; init=false, λ=T(1), proxg=NormL1(T(λ)),
                            kwargs...) where T                                       # src/lasso.jl:105-126
    p = fourier_problem(y, t, f, W)
    admm!(p, nothing, proxg, -1; kwargs...)               # Quadratic(Q, q=+A'Wy) as written (:119-121)
    params(p), f
end

function ls_sparse_spectral_lpv(y::AbstractVector{S}, X::AbstractVector{S}, V::AbstractVector{S}, w, Nv::Integer;
                                λ=1, coulomb=false, normalize=true, kwargs...) where S   # src/lasso.jl:27-70
    coulomb && throw(ArgumentError("coulomb=true is ill-defined in the sparse LPV path; use ls_spectral_lpv"))
    p = lpv_problem(y, X, V, w, Nv, normalize, false)
    local prm
    try
        admm!(p, nothing, GroupL2(λ, 2Nv), +1; kwargs...)
        prm = params(p, 0)
    catch e
        e isa InterruptException || rethrow(e)
        @info "Aborting"
        prm = params(p, 1)                                # z = copy(x)
    end
    SpectralExt(y, X, V, w[:], Nv, λ, coulomb, normalize, prm, nothing)
end

function ls_spectral_lpv(Y::AbstractVector, X::AbstractVector, V::AbstractVector, w, Nv::Integer;
                         λ=1e-8, coulomb=false, normalize=true)                      # src/lsfft.jl:239-259
    p = lpv_problem(Y, X, V, w, Nv, normalize, coulomb)
    SpectralExt(Y, X, V, w[:], Nv, λ, coulomb, normalize, pack(p, solve_ridge(p, λ^2)), nothing)
end

# ---- windows: src/windows.jl:27-42 (offsets from the C-ABI, views into y/t) ----------------------
struct Windows2; y; t; n::Int; noverlap::Int; W; offsets::Vector{Int64}; end
function Windows2(y::AbstractArray{T}, t, n::Int=length(y) >> 3, noverlap::Int=n >> 1, window_func=n -> ones(n)) where T
    noverlap < 0 && (noverlap = n >> 1)
    @assert length(y) == length(t) "y and t has to be the same length"
    k = Ref{Int64}(0)
    check(@ccall LIB.lpvs_window_count(length(y)::Int64, n::Int64, noverlap::Int64, k::Ref{Int64})::Int32)
    off = zeros(Int64, max(k[], 1))
    check(@ccall LIB.lpvs_window_offsets(length(y)::Int64, n::Int64, noverlap::Int64, off::Ptr{Int64}, length(off)::Int64, k::Ref{Int64})::Int32)
    Windows2(y, t, n, noverlap, T.(window_func(n)), off[1:k[]])
end
Base.length(w::Windows2) = length(w.offsets)
Base.iterate(w::Windows2, s=1) = s > length(w) ? nothing :
    ((view(w.y, w.offsets[s]+1:w.offsets[s]+w.n), view(w.t, w.offsets[s]+1:w.offsets[s]+w.n)), s + 1)

function ls_windowpsd(y, t, freqs=nothing; nw=8, noverlap=-1, window_func=n -> ones(n), estimator=ls_spectral, kwargs...)
    n = length(y) ÷ nw                                                               # src/lsfft.jl:112-126
    freqs === nothing && (freqs = default_freqs(t, n))
    windows = Windows2(y, t, n, noverlap, window_func)
    nw = length(windows)
    S = zeros(eltype(y), length(freqs))
    for (yi, ti) in windows
        x = estimator(yi, ti, freqs, windows.W; kwargs...)[1]
        S .+= abs2.(x)
    end
    S ./ nw^2, freqs
end

end # module
