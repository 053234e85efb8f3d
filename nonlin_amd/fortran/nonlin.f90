! Umbrella module: `use nonlin` keeps working for the hot-path types (src/nonlin.f90).
module nonlin
    use nonlin_types
    use nonlin_error_handling
    use nonlin_multi_eqn_mult_var
    use nonlin_linesearch
    use nonlin_solve
    use nonlin_least_squares
    use nonlin_polynomials
    use nonlin_multi_var
    use nonlin_optimize
    implicit none
    public
end module
