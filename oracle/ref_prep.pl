#!/usr/bin/perl
# oracle/ref_prep.pl REFDIR OUTDIR -- build step of oracle/_ref (TEST INFRASTRUCTURE ONLY).
# Rewrites the CUDA launch statements of the reference's kernel files so that g++ can compile the files, host entry
# points included, on top of ref_stubs/emul/cuda_emul.h.  The ONLY edit is
#     kernel<targs> <<< grid, block [, shmem] >>> (args);   ->   emul_launch(emul_cfg(grid, block [, shmem]), [&] { kernel<targs>(args); });
# Output goes to OUTDIR (a temp directory outside the repo, deleted by the Makefile after compiling): reference sources
# are never copied into the repository.
use strict; use warnings;
my ($ref, $out) = @ARGV;
my %expect = ('optimize_depth.cu' => 12, 'align_frame.cu' => 4, 'gblur.cu' => 2, 'fb_smooth.h' => 6, 'collect_p3p_instances.cu' => 2, 'meanshift.cu' => 2,
              'fit_robust_gaussian.cu' => 2, 'solve_batch_ap3p.cu' => 2, 'solve_batch_lambdatwist.cu' => 2);
for my $f (sort keys %expect) {
    open(my $in, '<', "$ref/gpu-kernels/$f") or die "$f: $!";
    local $/; my $src = <$in>; close $in;
    my $n = ($src =~ s/(\b[A-Za-z_]\w*\s*(?:<\s*\w+\s*>)?)\s*<<\s*<\s*(.*?)>>\s*>\s*\((.*?)\)\s*;/emul_launch(emul_cfg($2), [&] { $1($3); });/gs);
    $n ||= 0;
    die "$f: rewrote $n launches, expected $expect{$f} (reference changed?)\n" unless $n == $expect{$f};
    open(my $o, '>', "$out/$f") or die "$out/$f: $!";
    print $o $src; close $o;
}
